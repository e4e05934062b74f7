import numpy as np


def make_input(seed, b, t, h, w):
    """Same generator as tests/golden/make_golden_logits.py."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((b, t, 3, h, w)).astype(np.float32)


def assert_close(got, want, rtol, atol_scale=1e-4, what=''):
    """|got - want| <= rtol * |want| + atol_scale * max|want|  (elementwise), fp32 tolerance."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f'{what}: shape {got.shape} vs {want.shape}'
    assert np.isfinite(got).all(), f'{what}: non-finite values'
    scale = float(np.abs(want).max()) if want.size else 0.0
    err = np.abs(got - want)
    bound = rtol * np.abs(want) + atol_scale * scale
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(f'{what}: {int(bad.sum())}/{bad.size} outside tolerance; worst at {i}: '
                             f'got {got[i]:.7g} want {want[i]:.7g} err {err[i]:.3g} (scale {scale:.3g})')
    return float(err.max() / (scale + 1e-30))
