"""Eager launches vs a captured hipGraph of the same one-clip forward (torch.cuda.CUDAGraph around tsm_forward on the
capturing stream).  python tools/graph_probe.py [batch] [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f32'
eng = TsmEngine(max_clips=B, state_dict=make_state_dict(0, 12), dtype=dtype).warmup([B])
x = torch.randn(B, 8, 3, 224, 224, device='cuda')
out = torch.empty(B, 12, device='cuda')


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    back_to_back = (time.perf_counter() - t0) / n
    lat = []
    for _ in range(50):                       # latency of ONE forward from an idle stream, host-timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    return back_to_back * 1e3, sorted(lat)[25] * 1e3


eager = timeit(lambda: eng.forward_device(x, out=out))
ref = out.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.forward_device(x, out=out)
    torch.cuda.current_stream().synchronize()
    with torch.cuda.graph(g, stream=s):
        eng.forward_device(x, out=out)
torch.cuda.current_stream().wait_stream(s)
out.zero_()
g.replay()
torch.cuda.synchronize()
same = torch.equal(out, ref)
graph = timeit(g.replay)
print(f'batch {B} {dtype}: eager {eager[0]:.3f} ms back to back, {eager[1]:.3f} ms single-shot latency | '
      f'hipGraph replay {graph[0]:.3f} ms back to back, {graph[1]:.3f} ms single-shot | bitwise equal {same}')
eng.close()
