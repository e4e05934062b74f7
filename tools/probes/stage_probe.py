import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from workoutdetector_amd import inference_count as ic
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
print('torch threads', torch.get_num_threads())
pool = torch.randint(0, 256, (2616 + 8, 360, 206, 3), dtype=torch.uint8)
eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12), dtype='bf16x3', device=0)
side = torch.cuda.Stream()
for n in (2616, 2616, 800, 800, 800, 2616):
    vid = pool[:n]
    t0 = time.perf_counter()
    st = ic.stage_video(eng, vid, None, side)
    t1 = time.perf_counter()
    st.ready.synchronize()
    t2 = time.perf_counter()
    print(f'{n} frames: stage {1e3*(t1-t0):.1f} ms (+ copy wait {1e3*(t2-t1):.1f} ms), bytes {st.frames.numel()/1e6:.0f} MB')
# pieces
even = pool[0::2][:1308]
flat = torch.empty(even.numel(), dtype=torch.uint8, pin_memory=True)
dst = flat.view(even.shape)
for _ in range(3):
    t0 = time.perf_counter(); dst.copy_(even); t1 = time.perf_counter()
    print(f'copy_ strided->pinned {1e3*(t1-t0):.1f} ms = {even.numel()/1e9/(t1-t0):.1f} GB/s')
import numpy as np
for th in (1, 2, 4, 8):
    from concurrent.futures import ThreadPoolExecutor
    ex = ThreadPoolExecutor(th)
    n = even.shape[0]
    cuts = [(i * n // th, (i + 1) * n // th) for i in range(th)]
    torch.set_num_threads(1)
    t0 = time.perf_counter(); list(ex.map(lambda ab: dst[ab[0]:ab[1]].copy_(even[ab[0]:ab[1]]), cuts)); t1 = time.perf_counter()
    print(f'{th} threads x 1 torch thread: {1e3*(t1-t0):.1f} ms = {even.numel()/1e9/(t1-t0):.1f} GB/s')
torch.set_num_threads(16)
t0 = time.perf_counter(); d = dst.to('cuda', non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'H2D {1e3*(t1-t0):.1f} ms = {even.numel()/1e9/(t1-t0):.1f} GB/s')
t0 = time.perf_counter(); x = torch.empty(300 << 20, dtype=torch.uint8, pin_memory=True); t1 = time.perf_counter()
print(f'pin 300 MB: {1e3*(t1-t0):.1f} ms')
