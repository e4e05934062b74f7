"""PCIe-inclusive rate of the host-buffer boundary (TSM_MEM_HOST): clips/s of TsmEngine.run on numpy input.
    python tools/host_rate.py [dtype]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12), dtype=dtype).warmup([32])
x = np.random.default_rng(0).standard_normal((32, 8, 3, 224, 224), dtype=np.float32)
for _ in range(3):
    eng.run(None, {'input': x})
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    eng.run(None, {'input': x})
    ts.append(time.perf_counter() - t0)
med = sorted(ts)[len(ts) // 2]
print(f'{dtype}: host fp32 [32,8,3,224,224] (154 MB, pageable) -> logits on the host: {1e3 * med:.1f} ms per call = {32 / med:.0f} clips/s')
eng.close()
