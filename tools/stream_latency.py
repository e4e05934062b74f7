"""Latency of ONE live stream through StreamBatcher: from the arrival of a window's 8th frame to its state on the host
(360x206 uint8 frames like RepCount's stu* clips).  python tools/stream_latency.py [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.streaming import StreamBatcher  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12), dtype=dtype).warmup([1])
frames = np.random.default_rng(0).integers(0, 256, size=(8 * 60, 360, 206, 3), dtype=np.uint8)
sb = StreamBatcher(eng, max_batch=32)
lat, fwd = [], []
for w in range(60):
    for f in frames[w * 8:w * 8 + 7]:
        sb.push('solo', f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sb.push('solo', frames[w * 8 + 7])        # the 8th frame arrives: copy into the pinned window + step
    sb.step()
    lat.append(time.perf_counter() - t0)
    fwd.append(eng.last_forward_ms)
lat = sorted(lat[10:])
print(f'{dtype}: window latency (8th frame -> state) median {1e3 * lat[len(lat) // 2]:.3f} ms, min {1e3 * lat[0]:.3f} ms; '
      f'engine forward (HIP events) median {sorted(fwd[10:])[25]:.3f} ms')
eng.close()
