import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
sd = make_state_dict(0, 12)
B, T, S = 64, 16, 256
x = torch.randn(B, T, 3, S, S, device='cuda')
eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd, dtype='bf16')
eng.warmup([B])
out = torch.empty(B, 12, device='cuda')
for _ in range(3): eng.forward_device(x, out=out)
torch.cuda.synchronize()
eng.set_layer_timing(8)
for _ in range(8): eng.forward_device(x, out=out)
torch.cuda.synchronize()
lt = [eng.layer_times_ms(i) for i in range(8)]
print('TSM_STEM_2WG=%s stem us: %s  forward ms %.3f' % (os.environ.get('TSM_STEM_2WG'), sorted(round(d['conv1'] * 1e3, 1) for d in lt), eng.last_forward_ms))
