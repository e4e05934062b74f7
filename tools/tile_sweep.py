"""Per-layer kernel time for each forced conv tile shape (TSM_CONV_TILE), batch 32: which shape wins where.
Run on the GPU box:  python tools/tile_sweep.py > gpurun_out/tile_sweep.txt
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch, os
sys.path.insert(0, %r)
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
B = int(os.environ.get('TSM_SWEEP_BATCH', '32'))
eng = TsmEngine(max_clips=B, state_dict=make_state_dict(0, 12), dtype=os.environ.get('TSM_SWEEP_DTYPE', 'f32'))
x = torch.randn(B, 8, 3, 224, 224, device='cuda')
for _ in range(3): eng.forward_device(x)
torch.cuda.synchronize()
eng.set_layer_timing(8)
for _ in range(8): eng.forward_device(x)
torch.cuda.synchronize()
acc = {}
for i in range(8):
    for k, v in eng.layer_times_ms(i).items(): acc.setdefault(k, []).append(v)
print(json.dumps({k: sorted(v)[len(v)//2] for k, v in acc.items()}))
''' % ROOT

res = {}
for tile in os.environ.get('TSM_SWEEP_TILES', 'default,128x128,128x64,64x64').split(','):
    env = dict(os.environ)
    env['TSM_AUTOTUNE'] = '1' if tile == 'default' else '0'
    if len(sys.argv) > 1:
        env['TSM_SWEEP_DTYPE'] = sys.argv[1]
    if tile.startswith('code'):
        env['TSM_CONV_CODE'] = tile[4:]
        env['TSM_AUTOTUNE'] = '1'
    elif tile != 'default':
        env['TSM_CONV_TILE'] = tile
    out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        print(tile, 'FAILED', out.stderr[-2000:])
        continue
    res[tile] = json.loads(out.stdout.strip().splitlines()[-1])
names = list(res['default'])
print(f"{'layer':24s}" + ''.join(f'{t:>10s}' for t in res))
for n in names:
    print(f'{n:24s}' + ''.join(f'{res[t][n] * 1e3:10.1f}' for t in res))
print(f"{'total ms':24s}" + ''.join(f'{sum(max(0.0, v) for v in res[t].values()):10.3f}' for t in res))
best = sum(max(0.0, min(res[t][n] for t in res)) for n in names)
print('best-of per layer total ms:', round(best, 3))
