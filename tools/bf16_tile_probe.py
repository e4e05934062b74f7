"""Time the bf16 conv tiles on the config-5 layer shapes through the per-op C entry point's kernels (engine-level:
one engine per forced tile, per-launch HIP events).  python tools/bf16_tile_probe.py [batch] [segments] [size]"""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.flops import layer_table  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sd = make_state_dict(0, 12)
x = torch.randn(B, T, 3, S, S, device='cuda')
macs = {r['name']: r['macs'] for r in layer_table(S, S)}
res = {}
for tile in ('auto', '128x128', '256x256', '256x256p', 'ws'):
    os.environ.pop('TSM_CONV_TILE', None)
    os.environ.pop('TSM_AUTOTUNE', None)
    if tile != 'auto':
        os.environ['TSM_CONV_TILE'] = tile
        os.environ['TSM_AUTOTUNE'] = '0'
    eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd, dtype='bf16')
    eng.warmup([B])
    out = torch.empty(B, 12, device='cuda')
    for _ in range(3):
        eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    eng.set_layer_timing(8)
    for _ in range(8):
        eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    lt = [eng.layer_times_ms(i) for i in range(8)]
    names = eng.launch_names()
    res[tile] = ({k: sorted(d[k] for d in lt)[len(lt) // 2] for k in names}, eng.conv_tiles(B), out.cpu(), eng.last_forward_ms)
    eng.close()
print(f'batch {B}, T {T}, {S}x{S}, bf16: per-launch median us (TF/s)')
for k in res['auto'][0]:
    if '.conv' not in k and k != 'conv1':
        continue
    row = f'{k:26s}'
    for tile in res:
        us = res[tile][0][k] * 1e3
        tf = 2 * macs.get(k, 0) * B * T / (us * 1e-6) / 1e12 if us > 0 and k in macs else 0
        row += f' {tile}: {us:8.1f} ({tf:6.0f})' + (f' [{res[tile][1].get(k, "")}]' if tile == 'auto' else '')
    print(row)
for tile in res:
    print(tile, 'forward ms', round(res[tile][3], 3), 'bitwise == auto:', torch.equal(res[tile][2], res['auto'][2]))
