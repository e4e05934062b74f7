"""Per-launch times of the conv3 + downsample launches (layerN.0.conv3, conv_bf16_256p_kernel<1, false, false, true>) and the other
256-tile launches inside whole forwards of the bf16 engine at the config-5 size: python tools/dual_probe.py [clips] [pattern ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from workoutdetector_amd.engine import TsmEngine          # noqa: E402
from workoutdetector_amd.weights import make_state_dict    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pats = sys.argv[2:] or ['.0.conv3']
T, S = 16, 256
sd = make_state_dict(0, 12)
x = torch.randn(B, T, 3, S, S, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd, dtype='bf16')
eng.warmup([B])
out = torch.empty(B, 12, device='cuda')
for _ in range(3):
    eng.forward_device(x, out=out)
eng.set_layer_timing(8)
ms = []
for _ in range(8):
    eng.forward_device(x, out=out)
    ms.append(eng.last_forward_ms)
per = [eng.layer_times_ms(i) for i in range(8)]
names = [k for k in per[0] if any(p in k for p in pats)]
med = {k: sorted(p[k] for p in per)[4] for k in names}
print(f'forward {sorted(ms)[4]:.3f} ms; ' + ' '.join(f'{k[5:]}={v * 1e3:.0f}' for k, v in med.items() if v > 0) +
      f' ; sum {sum(med.values()) * 1e3:.0f} us; checksum {float(out.double().abs().sum()):.6f}', flush=True)
eng.close()
