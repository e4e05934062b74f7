"""K8: the fused HIP test transform against the CPU oracle (torchvision-0.13 tensor semantics)."""
import numpy as np
import pytest
import torch

from oracle import transform_oracle
from tests._stub import synthetic_video

pytestmark = pytest.mark.gpu


def _check(got, want, what):
    got, want = got.double(), want.double()
    err = (got - want).abs()
    bound = 1e-5 * want.abs() + 5e-4       # values reach ~1100 without /255: a few fp32 ulps of slack
    assert bool((err <= bound).all()), f'{what}: max err {float(err.max()):.3g}'


@pytest.mark.parametrize('h,w', [(360, 206), (272, 480), (256, 256), (224, 224), (720, 1280), (225, 640), (90, 52)])
@pytest.mark.parametrize('scale_255', [False, True])
def test_preprocess_matches_oracle(hip_lib, h, w, scale_255):
    from workoutdetector_amd.engine import preprocess_frames
    vid = torch.from_numpy(synthetic_video(h + w, 3, h, w))
    rng = np.random.default_rng(h * w)
    vid = torch.from_numpy(rng.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8))
    want = transform_oracle.test_transform(vid.permute(0, 3, 1, 2).float(), scale_255=scale_255)
    packed = preprocess_frames(vid.cuda(), scale_255=scale_255).cpu()
    assert tuple(packed.shape) == (3, 224, 224, 4) and float(packed[..., 3].abs().max()) == 0.0
    _check(packed[..., :3].permute(0, 3, 1, 2), want, 'u8 packed')
    nchw = preprocess_frames(vid.float().cuda(), scale_255=scale_255, packed=False).cpu()
    _check(nchw, want, 'f32 nchw')
    _check(nchw, packed[..., :3].permute(0, 3, 1, 2), 'u8 vs f32 source')  # separate instantiations: fma contraction may differ


def test_packed_layout_feeds_engine_like_nchw(hip_lib, sd0):
    """tsm_preprocess -> TSM_LAYOUT_NTHWC4 -> tsm_forward  ==  same frames handed over as NTCHW."""
    from workoutdetector_amd import _lib
    from workoutdetector_amd.engine import TsmEngine, preprocess_frames
    eng = TsmEngine(max_clips=2, state_dict=sd0)
    vid = torch.from_numpy(synthetic_video(3, 16, 120, 90)).cuda()
    packed = preprocess_frames(vid)                                     # [16,224,224,4]
    nchw = preprocess_frames(vid, packed=False)                         # [16,3,224,224]
    a = eng.forward_device(packed.reshape(2, 8, 224, 224, 4), layout=_lib.LAYOUT_NTHWC4).cpu()
    b = eng.forward_device(nchw.reshape(2, 8, 3, 224, 224)).cpu()
    assert torch.equal(a, b)
    from workoutdetector_amd._lib import TsmError
    with pytest.raises(TsmError):                                        # packed layout is device-only
        eng.forward_host(packed.cpu().numpy().reshape(2, 8, 224, 224, 4), layout=_lib.LAYOUT_NTHWC4)
    eng.close()


@pytest.mark.parametrize('crop,resize', [(224, 256), (33, 40)])
def test_pixel_pair_layouts_of_the_bf16_formats(hip_lib, crop, resize):
    """TSM_LAYOUT_NTHWC8B / NTHWC8S: one 8-element group per pixel PAIR (pixel 2j: c0 c1 c2 0, pixel 2j+1: ...),
    rows of ceil(crop/2) pairs, an odd crop ending in a zero pixel.  Decoded and compared with the fp32 NCHW
    output of the same kernel: bf16 = fp32 rounded to 8 significand bits (RNE); split = hi + lo within 2^-16."""
    from workoutdetector_amd import _lib
    from workoutdetector_amd.engine import preprocess_frames
    rng = np.random.default_rng(crop)
    vid = torch.from_numpy(rng.integers(0, 256, size=(2, 90, 52, 3), dtype=np.uint8)).cuda()
    want = preprocess_frames(vid, resize=resize, crop=crop, packed=False).cpu().permute(0, 2, 3, 1)   # [n,crop,crop,3]
    pairs = (crop + 1) // 2
    b = preprocess_frames(vid, resize=resize, crop=crop, layout=_lib.LAYOUT_NTHWC8B).cpu()
    assert tuple(b.shape) == (2, crop, pairs, 4)
    px = b.view(torch.bfloat16).reshape(2, crop, pairs * 2, 4).float()
    assert torch.equal(px[:, :, :crop, :3], want.to(torch.bfloat16).float())
    assert float(px[..., 3].abs().max()) == 0.0 and float(px[:, :, crop:].abs().max() if crop % 2 else 0.0) == 0.0
    s = preprocess_frames(vid, resize=resize, crop=crop, layout=_lib.LAYOUT_NTHWC8S).cpu()
    assert tuple(s.shape) == (2, crop, pairs, 8)
    g = s.view(torch.bfloat16).reshape(2, crop, pairs, 2, 8).float()       # [hi x8 | lo x8] per pair
    val = (g[:, :, :, 0] + g[:, :, :, 1]).reshape(2, crop, pairs * 2, 4)
    assert torch.equal(g[:, :, :, 0].reshape(2, crop, pairs * 2, 4)[:, :, :crop, :3], want.to(torch.bfloat16).float())
    err = (val[:, :, :crop, :3] - want).abs()
    assert bool((err <= want.abs() * 2.0 ** -16 + 1e-30).all())
