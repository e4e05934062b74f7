import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from tests._util import make_input
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
h = int(os.environ.get('S', '256')); t = int(os.environ.get('T', '16')); b = 2
sd = make_state_dict(11, 12)
x = make_input(500 + h + t, b, t, h, h)
got = {}
for flag in ('1', '0'):
    os.environ['TSM_FUSE_BLOCK'] = flag
    eng = TsmEngine(num_segments=t, height=h, width=h, max_clips=b, state_dict=sd, dtype='bf16')
    got[flag] = eng.forward_tap(x, 'layer1.1')
    eng.close()
a, c = got['1'], got['0']
bad = a != c
print('shape', a.shape, 'mismatch frac', bad.mean())
print('per frame', bad.reshape(a.shape[0], -1).mean(1).round(3))
print('per row (frame 1)', bad[1].reshape(a.shape[1], -1).mean(1).round(2))
print('per col (frame 1)', bad[1].transpose(1, 0, 2).reshape(a.shape[2], -1).mean(1).round(2))
print('per channel/8 (frame 1)', bad[1].reshape(-1, 32, 8).mean((0, 2)).round(2))
i = np.argwhere(bad)[:5]
for ix in i:
    print(tuple(ix), a[tuple(ix)], c[tuple(ix)])
print('finite', np.isfinite(a).mean(), 'absmax', np.nanmax(np.abs(a[np.isfinite(a)])))
