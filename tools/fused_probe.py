import os, sys, time, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
sd = make_state_dict(0, 12)
B = int(os.environ.get('B', '32'))
DTYPE = os.environ.get('DTYPE', 'f32')
T = int(os.environ.get('T', '8'))
S = int(os.environ.get('S', '224'))
x = torch.randn(B, T, 3, S, S, device='cuda')
res = {}
FLAGS = ('1',) if os.environ.get('ONLY') else ('0', '1', '')
for flag in FLAGS:
    if flag: os.environ['TSM_FUSE_CONV23'] = flag
    else: os.environ.pop('TSM_FUSE_CONV23', None)
    eng = TsmEngine(num_segments=T, height=S, width=S, max_clips=B, state_dict=sd, dtype=DTYPE)
    eng.warmup([B])
    out = torch.empty(B, 12, device='cuda')
    for _ in range(5): eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    eng.set_layer_timing(10)
    t0 = time.perf_counter()
    for _ in range(10): eng.forward_device(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    lt = [eng.layer_times_ms(i) for i in range(10)]
    names = eng.launch_names()
    avg = {k: sum(d[k] for d in lt) / 10 for k in names}
    tiles = eng.conv_tiles(B)
    print(f'fuse={flag or "auto"}: {dt*1e3:.3f} ms/forward  {B/dt:.1f} clips/s  (event-timed steps carry marker overhead)')
    for k in names:
        if k.startswith(('layer1.1', 'layer1.2', 'layer2.1', 'layer2.3')):
            print(f'   {k:22s} {avg[k]*1e3:8.1f} us  {tiles.get(k, "")}')
    res[flag or 'auto'] = out.cpu()
    eng.close()
if len(FLAGS) == 3:
    print('bitwise equal:', torch.equal(res['0'], res['1']), torch.equal(res['0'], res['auto']))
