"""fp32 headline shape (32 clips x 8 x 224^2): forward time with and without the tail split (tile code bit 0x200) in the tuner's
candidate list, per-layer times of the launches that picked it: python tools/tail_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from workoutdetector_amd.engine import TsmEngine          # noqa: E402
from workoutdetector_amd.weights import make_state_dict    # noqa: E402

B, T, S = 32, 8, 224
sd = make_state_dict(0, 12)
x = torch.randn(B, T, 3, S, S, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
os.environ['TSM_TUNE_CACHE'] = 'off'
res = {}
for flag in ('0', '1', '0', '1'):
    os.environ['TSM_TAIL_SPLIT'] = flag
    eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd)
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
    tiles = eng.conv_tiles(B)
    picked = [k for k, v in tiles.items() if 'tailK' in v]
    med = {k: sorted(p[k] for p in per)[4] for k in per[0]}
    seg = [k for k in med if k.startswith(('layer3.', 'layer4.')) and k.endswith(('conv1', 'conv2'))]
    print(f'TSM_TAIL_SPLIT={flag}: forward {sorted(ms)[4]:.3f} ms = {B / sorted(ms)[4] * 1e3:.1f} clips/s; tailK picked by {len(picked)} layers: {picked}')
    print('   ' + ' '.join(f'{k[5:]}={med[k] * 1e3:.0f}' for k in seg), flush=True)
    res.setdefault(flag, []).append(out.clone())
    eng.close()
assert torch.equal(res['0'][0], res['1'][0]), 'tail split != whole-K'
print('bit-identical logits')
