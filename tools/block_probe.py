"""Whole-Bottleneck kernel (bneck_ws_kernel, bf16 layer1.1 / layer1.2) against the separate launches: per-launch times of the
layer1 blocks and the forward, TSM_FUSE_BLOCK = 0 | 1 | auto, bitwise comparison of the logits.

    B=64 T=16 S=256 python tools/block_probe.py        (config 5; defaults)     B=32 T=8 S=224 for the headline shape"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

sd = make_state_dict(0, 12)
B, T, S = int(os.environ.get('B', '64')), int(os.environ.get('T', '16')), int(os.environ.get('S', '256'))
x = torch.randn(B, T, 3, S, S, device='cuda')
res = {}
for flag in ('0', '1', ''):
    if flag:
        os.environ['TSM_FUSE_BLOCK'] = flag
    else:
        os.environ.pop('TSM_FUSE_BLOCK', None)
    eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd, dtype='bf16')
    eng.warmup([B])
    out = torch.empty(B, 12, device='cuda')
    for _ in range(5):
        eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    eng.set_layer_timing(10)
    for _ in range(10):
        eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    lt = [eng.layer_times_ms(i) for i in range(10)]
    names = eng.launch_names()
    avg = {k: sum(d[k] for d in lt) / 10 for k in names}
    tiles = eng.conv_tiles(B)
    print(f'TSM_FUSE_BLOCK={flag or "auto"}: {dt * 1e3:.3f} ms/forward  {B / dt:.1f} clips/s', flush=True)
    for k in names:
        if k.startswith('layer1.'):
            print(f'   {k:22s} {avg[k] * 1e3:8.1f} us  {tiles.get(k, "")}')
    res[flag or 'auto'] = out.cpu()
    eng.close()
print('bitwise equal:', torch.equal(res['0'], res['1']), torch.equal(res['0'], res['auto']))
