"""Per-phase cycle sums of conv_bf16_256p_kernel (a -DTSM_256P_STAMP=1|2 build: workgroup 0 prints them at the kernel's end).
    TSM_LIB_PATH=tools/probes/bin/libtsm_p256stamp.so python tools/probes/p256_stamp.py"""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ['TSM_AUTOTUNE'] = '0'
os.environ['TSM_CONV_TILE'] = '256x256p'
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

eng = TsmEngine(num_segments=16, height=256, width=256, max_clips=64, state_dict=make_state_dict(0, 12), dtype='bf16')
x = torch.randn(64, 16, 3, 256, 256, device='cuda')
out = torch.empty(64, 12, device='cuda')
for i in range(3):
    print(f'--- forward {i}', flush=True)
    eng.forward_device(x, out=out)
    torch.cuda.synchronize()
eng.close()
