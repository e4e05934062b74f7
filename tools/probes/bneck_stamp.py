"""Per-phase cycle sums of bneck_ws_kernel (a -DTSM_BNECK_STAMP=1 build: the four waves of workgroup 0 print them at the kernel's end).
    TSM_LIB_PATH=tools/probes/bin/libtsm_bnstamp.so python tools/probes/bneck_stamp.py"""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ['TSM_AUTOTUNE'] = '0'
os.environ['TSM_FUSE_BLOCK'] = '1'
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

eng = TsmEngine(num_segments=16, height=256, width=256, max_clips=64, state_dict=make_state_dict(0, 12), dtype='bf16')
x = torch.randn(64, 16, 3, 256, 256, device='cuda')
out = torch.empty(64, 12, device='cuda')
for i in range(2):
    print(f'--- forward {i}', flush=True)
    eng.forward_device(x, out=out)
    torch.cuda.synchronize()
eng.close()
