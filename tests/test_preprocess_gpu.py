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


@pytest.mark.gpu
@pytest.mark.parametrize('layout_name', ['f32', 'bf16x3', 'nchw'])
def test_gather_clips_equals_the_reference_windows(hip_lib, layout_name):
    """tsm_gather_clips == the reference loop's windows (utils/inference_count.py:411-414: vid[i:i + 16:2] for i in
    range(0, len(vid), 8), the tail padded) on transformed frames, bit for bit against torch.index_select over the same
    buffer: whole videos and clip sub-ranges whose buffer starts mid-video, totals that end inside a window, on a step
    boundary and one frame past it; and the host-side validation refuses every range that would read outside the buffer."""
    import torch
    from workoutdetector_amd import _lib, engine
    from workoutdetector_amd import inference_count as ic
    g = torch.Generator().manual_seed(5)
    layout = {'f32': _lib.LAYOUT_NTHWC4, 'bf16x3': _lib.LAYOUT_NTHWC8S, 'nchw': _lib.LAYOUT_NTCHW}[layout_name]
    for total, (lo, hi) in [(77, (0, 10)), (77, (3, 7)), (64, (0, 8)), (65, (5, 9)), (9, (0, 2)), (16, (1, 2)), (1, (0, 1))]:
        video = torch.randint(0, 256, (total, 36, 52, 3), dtype=torch.uint8, generator=g)
        starts = ic.clip_starts(total)
        assert hi <= len(starts)
        f_lo = starts[lo] // 2
        f_hi = min((starts[hi - 1] + 16) // 2, (total + 1) // 2)
        even = torch.cat([video[0::2][f_lo:f_hi], torch.zeros((1, 36, 52, 3), dtype=torch.uint8)]).cuda()
        frames = engine.preprocess_frames(even, resize=32, crop=24, layout=layout)
        src = 8 * torch.arange(lo, hi)[:, None] + 2 * torch.arange(8)[None, :]
        idx = torch.where(src < total, src // 2 - f_lo, torch.full_like(src, frames.shape[0] - 1))
        want = frames[idx.cuda()]
        got = engine.gather_clips(frames, f_lo, total, lo, hi - lo)
        assert got.shape == want.shape and torch.equal(got, want), (total, lo, hi)
        # into a slice of a larger buffer, as the batcher does
        buf = torch.full((3 + (hi - lo) + 2, 8) + tuple(frames.shape[1:]), -7.0, device='cuda')
        engine.gather_clips(frames, f_lo, total, lo, hi - lo, out=buf[3:3 + hi - lo])
        assert torch.equal(buf[3:3 + hi - lo], want) and bool((buf[:3] == -7).all()) and bool((buf[3 + hi - lo:] == -7).all())
    # a padded tail on 16-byte frames, and more rows than one launch cuts (65535: the binding splits the range) -- the
    # windows stay those of index_select
    for n_buf, n_clips in ((4200, 1050), (70000, 17000)):       # 8 400 rows with a padded last clip; 136 000 rows = three launches
        buf = torch.arange(n_buf * 4, dtype=torch.float32, device='cuda').reshape(n_buf, 4)
        total = 2 * (n_buf - 1)
        got = engine.gather_clips(buf, 0, total, 0, n_clips)
        src = 8 * torch.arange(n_clips)[:, None] + 2 * torch.arange(8)[None, :]
        idx = torch.where(src < total, src // 2, torch.full_like(src, n_buf - 1))
        assert torch.equal(got, buf[idx.cuda()]), (n_buf, n_clips)
    # refusals: nothing may be launched for a range that leaves the buffer
    frames = torch.zeros((6, 24, 24, 4), device='cuda')
    for kw in [dict(first_frame=0, total_frames=77, first_clip=0, n_clips=2),      # needs 12 even frames, 6 in the buffer
               dict(first_frame=4, total_frames=77, first_clip=0, n_clips=1),      # clip 0 starts before the buffer
               dict(first_frame=0, total_frames=8, first_clip=1, n_clips=1),       # clip 1 starts at frame 8 = past the end
               dict(first_frame=0, total_frames=9, first_clip=0, n_clips=1, pad_frame=6)]:   # pad frame outside
        with pytest.raises(_lib.TsmError):
            engine.gather_clips(frames, **kw)
