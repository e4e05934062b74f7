"""Per-launch times of a small-batch forward (default 1 clip), fp32: where the latency of the streaming leg goes.
python tools/small_batch_probe.py [batch] [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.flops import layer_table  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
eng = TsmEngine(max_clips=B, state_dict=make_state_dict(0, 12), dtype=dtype)
x = torch.randn(B, 8, 3, 224, 224, device='cuda')
out = torch.empty(B, 12, device='cuda')
eng.warmup([B])
for _ in range(20):
    eng.forward_device(x, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    eng.forward_device(x, out=out)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 200
ev = []
for _ in range(20):
    eng.forward_device(x, out=out)
    ev.append(eng.last_forward_ms)
eng.set_layer_timing(16)
for _ in range(16):
    eng.forward_device(x, out=out)
torch.cuda.synchronize()
lt = [eng.layer_times_ms(i) for i in range(16)]
names = eng.launch_names()
med = {k: sorted(d[k] for d in lt)[8] for k in names}
tiles = eng.conv_tiles(B)
macs = {r['name']: r['macs'] for r in layer_table(224, 224)}
print(f'batch {B} {dtype}: wall {wall * 1e3:.3f} ms/forward (back to back), event {sorted(ev)[10]:.3f} ms; '
      f'sum of per-launch medians {sum(v for v in med.values() if v > 0):.3f} ms over {sum(v > 0 for v in med.values())} timed launches')
for k in names:
    if med[k] <= 0:
        continue
    tf = 2 * macs.get(k, 0) * B * 8 / (med[k] * 1e-3) / 1e12 if k in macs else 0
    print(f'  {k:24s} {med[k] * 1e3:8.1f} us  {tf:6.1f} TF/s  {tiles.get(k, "")}')
eng.close()
