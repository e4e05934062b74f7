"""Print the conv tile / split-K choice of the autotuner per layer:  python tools/show_tiles.py [n_clips] [dtype]"""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
eng = TsmEngine(max_clips=n, state_dict=make_state_dict(0, 12), dtype=dtype)
x = torch.randn(n, 8, 3, 224, 224, device='cuda')
for _ in range(3):
    eng.forward_device(x)
torch.cuda.synchronize()
t = eng.conv_tiles(n)
print(Counter(t.values()))
for k, v in t.items():
    print(f'{k:28s} {v}')
