import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
dtype, B, T, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd = make_state_dict(0, 12)
x = torch.randn(B, T, 3, S, S, device='cuda')
ref = None
for zz in ('0', '1', '0', '1'):
    os.environ['TSM_ZIGZAG'] = zz
    eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd, dtype=dtype)
    eng.warmup([B])
    out = torch.empty(B, 12, device='cuda')
    for _ in range(3):
        eng.forward_device(x, out=out)
    ms = []
    for _ in range(12):
        eng.forward_device(x, out=out)
        ms.append(eng.last_forward_ms)
    ms.sort()
    o = out.cpu()
    ref = o if ref is None else ref
    print(f'{dtype} B={B} T={T} {S}^2 zigzag={zz}: forward median {ms[6]:.3f} ms min {ms[0]:.3f}  bitwise {torch.equal(o, ref)}', flush=True)
    eng.close()
