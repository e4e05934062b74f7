"""A/B of shift + conv1 + the stride-2 conv2 of layer2.0 as ONE launch (front_s2_kernel) against the two tuned launches, whole forwards
of the bf16 engine at the config-5 size: python tools/front_probe.py [clips]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from workoutdetector_amd.engine import TsmEngine          # noqa: E402
from workoutdetector_amd.weights import make_state_dict    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, S = 16, 256
sd = make_state_dict(0, 12)
x = torch.randn(B, T, 3, S, S, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
res = {}
for flag in ('0', '1', '0', '1', ''):
    if flag:
        os.environ['TSM_FUSE_FRONT'] = flag
    else:
        os.environ.pop('TSM_FUSE_FRONT', None)
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
    med = {k: sorted(p[k] for p in per)[4] for k in per[0] if k.startswith('layer2.0')}
    print(f'TSM_FUSE_FRONT={flag or "auto"}: forward {sorted(ms)[4]:.3f} ms; ' + ' '.join(f'{k[7:]}={v * 1e3:.0f}' for k, v in med.items() if v > 0) +
          f'  [{eng.conv_tiles(B)["layer2.0.conv1"]}]', flush=True)
    res.setdefault(flag or 'auto', []).append((sorted(ms)[4], out.clone()))
    eng.close()
assert torch.equal(res['0'][0][1], res['1'][0][1]) and torch.equal(res['0'][0][1], res['auto'][0][1]), 'fused != separate'
print('bit-identical logits; forward ms separate', [round(r[0], 3) for r in res['0']], 'fused', [round(r[0], 3) for r in res['1']])
