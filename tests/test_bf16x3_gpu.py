"""TSM_DTYPE_BF16X3 ("split-bf16": hi/lo bf16 storage, three bf16 MFMAs per product, fp32 accumulate)
against the fp32 CPU oracle.  Same bar as the exact-fp32 engine: fp32 rtol 1e-3 on logits and stage taps;
per-op bar 3e-4 (the format keeps ~17 significand bits per value, fp32 keeps 24)."""
import json

import numpy as np
import pytest
import torch

from oracle import transform_oracle, tsm_oracle
from tests._util import assert_close, make_input
from tests.test_ops_gpu import CONV_CASES, _bn, _nchw, _nhwc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,hi,wi,cin,cout,k,stride,relu,use_res,shiftT', CONV_CASES)
def test_conv_bn_act_x3(hip_lib, n, hi, wi, cin, cout, k, stride, relu, use_res, shiftT):
    from workoutdetector_amd.engine import conv_bn_act_nhwc
    g = torch.Generator().manual_seed(1000 + cin + cout + k + hi)
    x = torch.randn(n, cin, hi, wi, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    bn = _bn(cout, g)
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    xin = tsm_oracle.temporal_shift(x, shiftT, 8) if shiftT else x
    want = tsm_oracle.conv_bn_act(xin, w, bn, stride, pad, relu, res)
    got = conv_bn_act_nhwc(_nhwc(x).cuda(), w.cuda(), *[b.cuda() for b in bn], stride=stride, relu=relu,
                           residual=None if res is None else _nhwc(res).cuda(), shift_segments=shiftT, fold_div=8,
                           dtype='bf16x3')
    assert_close(_nchw(got.cpu()).numpy(), want.numpy(), rtol=3e-4, atol_scale=3e-4, what='conv x3')


@pytest.fixture(scope='module')
def engine_x3(hip_lib, sd0):
    from workoutdetector_amd.engine import TsmEngine
    eng = TsmEngine(max_clips=4, state_dict=sd0, dtype='bf16x3')
    yield eng
    eng.close()


def test_logits_and_taps_x3(engine_x3, sd0, capsys):
    x = make_input(100, 2, 8, 224, 224)
    taps = {}
    want = tsm_oracle.tsm_forward(sd0, torch.from_numpy(x), taps=taps).numpy()
    got = engine_x3.run(None, {'input': x})[0]
    worst = assert_close(got, want, rtol=1e-3, atol_scale=1e-5, what='logits bf16x3')
    with capsys.disabled():
        print(f'\n[bf16x3] logits max|err|/scale = {worst:.3g} (bar 1e-3)')
    for stage in ['stem', 'layer1.0', 'layer2.0', 'layer3.5', 'layer4.2']:
        got_t = engine_x3.forward_tap(x, stage)
        # 3e-5 of the tap's scale: the split format drops the lo*lo product (2^-16 per product), which on the few
        # near-zero elements of a 3-million-element tap shows as 1.2e-5 of the scale (measured); logits hold 1e-5
        worst_t = assert_close(got_t, taps[stage].permute(0, 2, 3, 1).numpy(), rtol=1e-3, atol_scale=3e-5, what=stage)
        with capsys.disabled():
            print(f'[bf16x3] {stage} max|err|/scale = {worst_t:.3g}')


def test_golden_logits_x3(hip_lib, golden_dir):
    from workoutdetector_amd.engine import TsmEngine
    from workoutdetector_amd.weights import make_state_dict
    gold = json.load(open(f'{golden_dir}/tsm_r50_logits.json'))
    for name, case in gold.items():
        b, t, _, h, w = case['shape']
        eng = TsmEngine(num_segments=t, height=h, width=w, max_clips=b, dtype='bf16x3',
                        state_dict=make_state_dict(case['weight_seed'], 12))
        got = eng.run(None, {'input': make_input(case['input_seed'], b, t, h, w)})[0]
        assert_close(got, np.array(case['logits'], dtype=np.float32), rtol=1e-3, atol_scale=1e-5, what=name)
        eng.close()


def test_x3_agrees_with_exact_f32_engine_and_is_batch_invariant(engine_x3, sd0):
    from workoutdetector_amd.engine import TsmEngine
    x = make_input(11, 5, 8, 224, 224)
    a = engine_x3.run(None, {'input': x})[0]                 # chunks of 4 + 1
    for i in (0, 4):
        assert np.array_equal(engine_x3.run(None, {'input': x[i:i + 1]})[0][0], a[i])
    f32 = TsmEngine(max_clips=4, state_dict=sd0)
    b = f32.run(None, {'input': x})[0]
    f32.close()
    assert_close(a, b, rtol=5e-4, atol_scale=1e-4, what='bf16x3 vs f32 engine')


def test_preprocess_split_layout_feeds_x3_engine(engine_x3):
    from workoutdetector_amd import _lib
    from workoutdetector_amd._lib import TsmError
    from workoutdetector_amd.engine import preprocess_frames
    from tests._stub import synthetic_video
    vid = torch.from_numpy(synthetic_video(3, 16, 120, 90))
    packed = preprocess_frames(vid.cuda(), layout=engine_x3.packed_layout)
    assert engine_x3.packed_layout == _lib.LAYOUT_NTHWC8S and tuple(packed.shape) == (16, 224, 112, 8)
    a = engine_x3.forward_device(packed.reshape(2, 8, 224, 112, 8), layout=_lib.LAYOUT_NTHWC8S).cpu()
    want_in = transform_oracle.test_transform(vid.permute(0, 3, 1, 2).float()).reshape(2, 8, 3, 224, 224)
    b = engine_x3.run(None, {'input': want_in.numpy()})[0]
    assert_close(a.numpy(), b, rtol=1e-4, atol_scale=1e-4, what='packed split input')
    with pytest.raises(TsmError):                                    # fp32-packed frames into a split engine
        engine_x3.forward_device(preprocess_frames(vid.cuda()).reshape(2, 8, 224, 224, 4), layout=_lib.LAYOUT_NTHWC4)
