"""Per-launch medians of the launches that run on conv_bf16_256_kernel (or, with the argument 256x256p, its persistent
form) at the config-5 size: one engine, the tile forced wherever it applies.
For A/B builds:  TSM_LIB_PATH=<variant.so> python tools/k256_probe.py [256x256|256x256p]"""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['TSM_CONV_TILE'] = sys.argv[1] if len(sys.argv) > 1 else '256x256'
os.environ['TSM_AUTOTUNE'] = '0'
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.flops import layer_table  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

B, T, S = 64, 16, 256
eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=make_state_dict(0, 12), dtype='bf16')
x = torch.randn(B, T, 3, S, S, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))   # (seeded: the checksum compares builds / tiles)
out = torch.empty(B, 12, device='cuda')
for _ in range(4):
    eng.forward_device(x, out=out)
torch.cuda.synchronize()
eng.set_layer_timing(12)
for _ in range(12):
    eng.forward_device(x, out=out)
torch.cuda.synchronize()
lt = [eng.layer_times_ms(i) for i in range(12)]
names = eng.launch_names()
med = {k: sorted(d[k] for d in lt)[6] for k in names}
macs = {r['name']: r['macs'] for r in layer_table(S, S)}
fw = []
for _ in range(8):
    eng.forward_device(x, out=out)
    fw.append(eng.last_forward_ms)
tot = 0.0
print('lib', os.environ.get('TSM_LIB_PATH', 'default'), 'tile', os.environ['TSM_CONV_TILE'])
for k in names:
    if k.startswith(('layer2', 'layer3', 'layer4')) and med[k] > 0:
        extra = macs.get(k.replace('conv3', 'downsample'), 0) if k.endswith('.0.conv3') else 0
        tf = 2 * (macs[k] + extra) * B * T / (med[k] * 1e-3) / 1e12
        tot += med[k]
        print(f'  {k:18s} {med[k] * 1e3:8.1f} us {tf:7.0f} TF/s')
print(f'  layer2-4 sum {tot:.3f} ms; forward median {sorted(fw)[4]:.3f} ms; checksum {float(out.double().sum()):.6f}')
eng.close()
