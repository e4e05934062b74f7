"""Per-kernel parity: HIP kernels (through the C ABI) vs the CPU oracle on seeded inputs.

fp32 tolerance for the conv tests: |hip - oracle| <= 1e-4*|oracle| + 1e-4*max|oracle| -- both sides
accumulate in fp32 but in different orders (MFMA k-chain vs oneDNN blocking), and the engine folds
the BatchNorm scale into the weights before the multiply.  Data-movement kernels are bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import tsm_oracle
from tests._util import assert_close

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize('nb,t,c,h,w,div', [(1, 8, 64, 5, 7, 8), (2, 8, 256, 3, 3, 8), (3, 4, 32, 2, 5, 8),
                                            (1, 16, 128, 4, 4, 8), (2, 1, 64, 2, 2, 8), (1, 8, 2048, 1, 1, 8),
                                            (2, 8, 64, 3, 3, 16)])
def test_temporal_shift_bit_exact(hip_lib, nb, t, c, h, w, div):
    from workoutdetector_amd.engine import temporal_shift_nhwc
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(nb * t, c, h, w, generator=g)
    want = tsm_oracle.temporal_shift(x, t, div)
    got = _nchw(temporal_shift_nhwc(_nhwc(x).cuda(), t, div).cpu())
    assert torch.equal(got, want)


def test_temporal_shift_reference_vectors(hip_lib, golden_dir):
    """Against outputs of the reference's own TemporalShift.shift (tests/golden/ref_temporal_shift.npz)."""
    from workoutdetector_amd.engine import temporal_shift_nhwc
    z = np.load(f'{golden_dir}/ref_temporal_shift.npz')
    ran = 0
    for i, (nb, t, c, h, w, div) in enumerate(z['meta']):
        if (c // div) % 4 or c % 4:
            continue  # kernel contract: fold % 4 == 0 (always true for ResNet-50 channel counts)
        x = torch.from_numpy(z[f'x{i}'])
        got = _nchw(temporal_shift_nhwc(_nhwc(x).cuda(), int(t), int(div)).cpu())
        assert torch.equal(got, torch.from_numpy(z[f'y{i}'])), i
        ran += 1
    assert ran >= 4


def _bn(c, g):
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1,
            torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5)


CONV_CASES = [
    # n, hi, wi, cin, cout, k, stride, relu, residual, shiftT
    (8, 14, 14, 64, 64, 1, 1, True, False, 8),      # layer1.0.conv1 shape class, fold = 8
    (8, 7, 9, 256, 64, 1, 1, True, False, 8),       # ragged M (504 rows), shift fold 32
    (4, 14, 14, 64, 256, 1, 1, True, True, 0),      # conv3 + residual + relu
    (4, 14, 14, 256, 512, 1, 2, False, False, 0),   # strided downsample, no relu
    (2, 15, 13, 512, 1024, 1, 2, False, False, 0),  # odd sizes, stride 2
    (4, 14, 14, 64, 64, 3, 1, True, False, 0),      # 3x3 s1
    (4, 14, 14, 128, 128, 3, 2, True, False, 0),    # 3x3 s2 (v1.5 stride on conv2)
    (3, 9, 11, 256, 256, 3, 1, True, False, 0),     # odd sizes, ragged M
    (2, 7, 7, 512, 512, 3, 1, True, False, 0),      # layer4 shape class, tiny M
    (2, 32, 32, 3, 64, 7, 2, True, False, 0),       # stem
    (3, 45, 37, 3, 64, 7, 2, True, False, 0),       # stem, odd sizes
    # >= 256 tiles of 128 rows -> the 128x128 / 128x64 tile paths (conv_tile_shape)
    (8, 48, 48, 64, 256, 3, 1, True, False, 0),     # 128x128 tiles, 3x3
    (32, 28, 28, 256, 256, 1, 1, True, False, 8),   # 128x128 tiles with shift
    (16, 48, 48, 64, 64, 3, 1, True, False, 0),     # 128x64 tiles
    (16, 28, 28, 128, 512, 1, 1, True, True, 0),    # 128x128 + residual
    (16, 47, 45, 64, 64, 1, 1, True, False, 8),     # 128x64 tiles, ragged M, shift fold 8
    (4, 96, 96, 3, 64, 7, 2, True, False, 0),       # stem on 128x64 tiles
]


@pytest.mark.parametrize('n,hi,wi,cin,cout,k,stride,relu,use_res,shiftT', CONV_CASES)
def test_conv_bn_act(hip_lib, n, hi, wi, cin, cout, k, stride, relu, use_res, shiftT):
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
                           residual=None if res is None else _nhwc(res).cuda(), shift_segments=shiftT, fold_div=8)
    assert_close(_nchw(got.cpu()).numpy(), want.numpy(), rtol=1e-4, atol_scale=1e-4, what='conv')


def test_conv_identity_asymmetric(hip_lib):
    """A = I style check with an asymmetric kernel: catches transposed C/D maps and swapped k order."""
    from workoutdetector_amd.engine import conv_bn_act_nhwc
    n, hi, wi, cin, cout = 2, 8, 8, 64, 128
    x = torch.zeros(n, cin, hi, wi)
    for c in range(cin):
        x[:, c, c % hi, (3 * c) % wi] = float(c + 1)
    w = torch.zeros(cout, cin, 1, 1)
    for o in range(cout):
        w[o, (5 * o + 1) % cin, 0, 0] = float(o + 1)
    ones, zeros = torch.ones(cout), torch.zeros(cout)
    bn = (ones, zeros, zeros, ones - 1e-5)
    want = tsm_oracle.conv_bn_act(x, w, bn, 1, 0, False)
    got = conv_bn_act_nhwc(_nhwc(x).cuda(), w.cuda(), *[b.cuda() for b in bn], stride=1, relu=False)
    assert_close(_nchw(got.cpu()).numpy(), want.numpy(), rtol=1e-6, atol_scale=0.0, what='identity conv')


@pytest.mark.parametrize('n,h,w,c', [(2, 16, 16, 64), (3, 15, 13, 64), (1, 112, 112, 64), (2, 5, 7, 8)])
def test_maxpool(hip_lib, n, h, w, c):
    from workoutdetector_amd.engine import maxpool3x3s2_nhwc
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(h))
    want = torch.nn.functional.max_pool2d(x, 3, 2, 1)
    got = _nchw(maxpool3x3s2_nhwc(_nhwc(x).cuda()).cpu())
    assert torch.equal(got, want)


@pytest.mark.parametrize('b,t,hw,cls', [(1, 8, 7, 12), (4, 8, 7, 12), (3, 16, 8, 5), (2, 1, 1, 7)])
def test_head(hip_lib, b, t, hw, cls):
    from workoutdetector_amd.engine import head_nhwc
    g = torch.Generator().manual_seed(b * 10 + t)
    feat = torch.randn(b * t, 2048, hw, hw, generator=g)
    sd = {'fc.weight': torch.randn(cls, 2048, generator=g) * 0.05, 'fc.bias': torch.randn(cls, generator=g)}
    want = tsm_oracle.head(feat, sd, t)
    got = head_nhwc(_nhwc(feat).cuda(), sd['fc.weight'].cuda(), sd['fc.bias'].cuda(), t).cpu()
    assert_close(got.numpy(), want.numpy(), rtol=1e-4, atol_scale=1e-5, what='head')


def test_head_reference_consensus_vectors(hip_lib, golden_dir):
    """Segment consensus pinned by the reference's own SegmentConsensus outputs: with hw = 1 and an
    identity classifier the head reduces to mean over segments."""
    from workoutdetector_amd.engine import head_nhwc
    z = np.load(f'{golden_dir}/ref_consensus.npz')
    for i in range(4):
        x = torch.from_numpy(z[f'x{i}'])           # [b, t, c]
        b, t, c = x.shape
        feat = torch.zeros(b * t, 1, 1, 2048)
        feat[:, 0, 0, :c] = x.reshape(b * t, c)
        w = torch.zeros(c, 2048)
        w[torch.arange(c), torch.arange(c)] = 1.0
        got = head_nhwc(feat.cuda(), w.cuda(), torch.zeros(c).cuda(), t).cpu()
        assert_close(got.numpy(), z[f'y{i}'], rtol=1e-6, atol_scale=1e-6, what=f'consensus{i}')


def test_scores_to_states_on_gpu_equals_the_host_path(hip_lib, golden_dir):
    """K9 (tsm_scores_to_states) against counting.scores_to_preds and the reference's executed to_softmax
    (tests/golden/ref_metrics.json): identical integer states, with and without softmax, at thresholds other than 0.5,
    ties -> first index, exactly-at-threshold kept (``>=``); probabilities within 2 ulp (expf is the only freedom)."""
    import json

    from oracle import counting_oracle
    from workoutdetector_amd.counting import scores_to_preds, softmax_rows
    from workoutdetector_amd.engine import scores_to_states
    ref = json.load(open(f'{golden_dir}/ref_metrics.json'))['to_softmax']
    rows = [[c['scores'][str(j)] for j in range(12)] for c in ref]
    rng = np.random.default_rng(9)
    for scale in (0.3, 1.0, 3.0, 10.0):
        rows += (rng.standard_normal((200, 12)) * scale).astype(np.float32).tolist()
    rows += [[0.0, 0.0] + [-1000.0] * 10,                      # two-way tie at p = 0.5 exactly: kept, first index
             [1.0] * 12,                                       # twelve-way tie: class 0, p = 1/12 < 0.5 -> -1
             [-100.0] * 11 + [5.0], [3.0, 3.0, 2.9] + [0.0] * 9]
    x = torch.tensor(rows, dtype=torch.float32).cuda()
    for softmax in (True, False):
        for thr in (0.5, 0.3, 0.9):
            st, top = scores_to_states(x, threshold=thr, softmax=softmax, return_top=True)
            got = st.cpu().tolist()
            assert got == scores_to_preds(rows, threshold=thr, softmax=softmax), (softmax, thr)
            assert got == counting_oracle.scores_to_preds(rows, threshold=thr, use_softmax=softmax)
            p = softmax_rows(np.asarray(rows, np.float32)) if softmax else np.asarray(rows, np.float32)
            np.testing.assert_allclose(top.cpu().numpy(), p.max(axis=1), rtol=3e-7, atol=0)
    # the reference's own softmax outputs: the winning probability agrees with the executed to_softmax
    st, top = scores_to_states(x[:len(ref)], return_top=True)
    want = np.float32([max(c['softmax'].values()) for c in ref])
    np.testing.assert_allclose(top.cpu().numpy(), want, rtol=3e-7)
    # other class counts (numpy's pairwise order only applies from 8 elements on)
    for c in (2, 5, 8, 17, 130):
        r = (rng.standard_normal((64, c)) * 2).astype(np.float32)
        assert scores_to_states(torch.from_numpy(r).cuda()).cpu().tolist() == scores_to_preds(r.tolist())
