import numpy as np


def make_input(seed, b, t, h, w):
    """Same generator as tests/golden/make_golden_logits.py."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((b, t, 3, h, w)).astype(np.float32)


# ---- "the kernel under test is the one that ran" ---------------------------------------------------------------------
# (workoutdetector_amd.engine.launch_trace records the library's kernel launches of the calling thread; a forced kernel
#  form that silently fell back would otherwise make every "bit-identical" comparison trivially true)
IGEMM_TILE_DIMS = {'64x64': (64, 64, 2, 2), '128x64': (128, 64, 2, 2), '128x128': (128, 128, 2, 2),
                   '128x128w8': (128, 128, 4, 2), '32x32': (32, 32, 1, 1)}
TILE_KERNELS = {'256x256': ('conv_bf16_256_kernel<',), '256x256p': ('conv_bf16_256p_kernel<',),
                'ws': ('conv3x3_ws_kernel<', 'conv3x3_ws128_kernel<', 'conv1x1_ws_kernel<', 'conv1x1_wsn_kernel<')}


def ran_tile(trace, tile):
    """Did a launch of the kernel family behind TSM_CONV_TILE=`tile` appear in the trace?"""
    if tile in IGEMM_TILE_DIMS:
        want = '[BM = %d, BN = %d, WGM = %d, WGN = %d,' % IGEMM_TILE_DIMS[tile]
        return any(k.startswith('conv_igemm<') and want in k for k in trace.kernels)
    return any(trace.ran(p) for p in TILE_KERNELS[tile])


def assert_ran(trace, prefix, what=''):
    assert trace.ran(prefix), f'{what}: no `{prefix}` launch; the trace holds {sorted(set(trace.kernels))}'


def assert_not_ran(trace, prefix, what=''):
    assert not trace.ran(prefix), f'{what}: `{prefix}` ran although it must not: {sorted(set(trace.kernels))}'


def assert_ran_tile(trace, tile, what=''):
    assert ran_tile(trace, tile), f'{what}: TSM_CONV_TILE={tile} did not run its kernel; the trace holds {sorted(set(trace.kernels))}'


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


BF16_OP_RTOL = 2.0 ** -8      # half a bf16 ulp (8 significand bits) relative to the value: 3.9e-3
BF16_E2E_BAR = 1e-2           # logits of the bf16 engine vs the bf16-storage oracle, fraction of the logit scale
                              # (measured on MI355X: 3e-4 .. 2.9e-3 over the suite's shapes; vs the fp32 oracle 2e-3 .. 5e-3)
BF16_TAP_BAR = 2e-2           # the same for the worst ELEMENT of a stage tap (millions of elements, up to 50 roundings
                              # deep: measured 3e-3 after layer1.0 rising to 1.2e-2 after layer4.2)


def assert_bf16_op(got, want_unrounded, what='', atol_scale=2e-5):
    """A bf16-storage kernel's output against the bf16 oracle's UNROUNDED fp32 result (operands rounded exactly as the
    kernel rounds them, fp32 accumulate): the stored value must be a correct rounding of it -- within half a bf16 ulp
    (rtol 2^-8 <= 4e-3) plus the fp32 accumulation-order noise (atol_scale of the tensor's scale)."""
    return assert_close(got, want_unrounded, rtol=BF16_OP_RTOL, atol_scale=atol_scale, what=what)


def bf16_logits_report(got, want_bf16, want_f32, what, capsys=None):
    """The bf16 engine's logits: asserted against the bf16-storage oracle (BF16_E2E_BAR of the scale, same arg-max),
    REPORTED against the fp32 oracle (the accuracy figure of the mode, not a parity bar)."""
    got, want_bf16, want_f32 = (np.asarray(a, dtype=np.float64) for a in (got, want_bf16, want_f32))
    scale = float(np.abs(want_f32).max())
    e_model = float(np.abs(got - want_bf16).max()) / scale
    e_f32 = float(np.abs(got - want_f32).max()) / scale
    msg = f'[{what}] logits max|err|/scale: {e_model:.3g} vs the bf16-storage oracle (bar {BF16_E2E_BAR:g}), {e_f32:.3g} vs the fp32 oracle'
    if capsys is not None:
        with capsys.disabled():
            print('\n' + msg)
    assert np.isfinite(got).all() and e_model <= BF16_E2E_BAR, msg
    assert (got.argmax(1) == want_bf16.argmax(1)).all(), msg
    return e_model, e_f32


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
