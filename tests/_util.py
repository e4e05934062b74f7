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


def fit_probe_fc(sd, video_u8, num_class=12, on=4, off=5, gain=8.0):
    """Classifier for the seeded trunk that separates bright from dark clips of a synthetic stream, so
    that random-init weights give alternating start/end states (classes ``on``/``off``) and a
    non-trivial repetition count.  Uses the ORACLE trunk on the CPU (test infrastructure only)."""
    import torch
    from oracle import transform_oracle, tsm_oracle
    feats, level = [], []
    for s in range(0, video_u8.shape[0], 8):
        clip = transform_oracle.make_clip(video_u8, s)
        x = transform_oracle.clip_to_input(clip)[0]
        f = tsm_oracle.trunk(x, sd, 8)
        feats.append(torch.nn.functional.adaptive_avg_pool2d(f, 1).flatten(1).mean(0))
        level.append(float(clip.mean()))
    feats = torch.stack(feats)
    level = torch.tensor(level)
    bright = level > level.median()
    d = feats[bright].mean(0) - feats[~bright].mean(0)
    mid = 0.5 * (feats[bright].mean(0) + feats[~bright].mean(0))
    w = torch.zeros(num_class, feats.shape[1])
    b = torch.full((num_class,), -3.0 * gain)
    scale = gain / float(d @ d) * 2.0
    w[on], w[off] = d * scale, -d * scale
    b[on], b[off] = -float(mid @ d) * scale, float(mid @ d) * scale
    return w, b
