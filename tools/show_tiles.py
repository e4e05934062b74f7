import sys, torch
sys.path.insert(0, '/root/repo')
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12))
x = torch.randn(32, 8, 3, 224, 224, device='cuda')
for _ in range(3): eng.forward_device(x)
torch.cuda.synchronize()
t = eng.conv_tiles(32)
from collections import Counter
print(Counter(t.values()))
print({k: v for k, v in t.items() if 'layer4' in k or 'layer3.1' in k})
