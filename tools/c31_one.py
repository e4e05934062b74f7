"""Per-launch times of the cross-block kernel's sites at the config-5 size (one engine, TSM_FUSE_C3C1 from the environment)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from workoutdetector_amd.engine import TsmEngine          # noqa: E402
from workoutdetector_amd.weights import make_state_dict    # noqa: E402

B, T, S = 64, 16, 256
eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=make_state_dict(0, 12), dtype='bf16')
x = torch.randn(B, T, 3, S, S, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
eng.warmup([B])
out = torch.empty(B, 12, device='cuda')
for _ in range(2):
    eng.forward_device(x, out=out)
eng.set_layer_timing(6)
ms = []
for _ in range(6):
    eng.forward_device(x, out=out)
    ms.append(eng.last_forward_ms)
per = [eng.layer_times_ms(i) for i in range(6)]
keys = ['layer2.1.conv3', 'layer2.2.conv3', 'layer2.3.conv3', 'layer3.1.conv3', 'layer3.2.conv3', 'layer3.4.conv3']
print(f'forward {sorted(ms)[3]:.3f} ms; ' + ' '.join(f'{k[5:]}={sorted(p[k] for p in per)[3] * 1e3:.0f}' for k in keys), flush=True)
eng.close()
