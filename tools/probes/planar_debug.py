import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
from workoutdetector_amd.engine import TsmEngine
from workoutdetector_amd.weights import make_state_dict
sd = make_state_dict(0, 12)
h = w = 64
x = np.random.default_rng(1).standard_normal((1, 8, 3, h, w)).astype(np.float32)
out = {}
for dtype in ('bf16', 'f32'):
    for flag in ('0', '1'):
        os.environ['TSM_STEM_PLANAR'] = flag
        eng = TsmEngine(height=h, width=w, max_clips=1, state_dict=sd, dtype=dtype)
        out[flag] = eng.forward_tap(x, 'stem')
        eng.close()
    a, b = out['0'], out['1']
    d = a != b
    print(dtype, 'shape', a.shape, 'differ', d.sum(), 'of', d.size)
    if d.any():
        idx = np.argwhere(d)
        print(' first', idx[:5].tolist(), 'frames', np.unique(idx[:, 0]).tolist(), 'rows', np.unique(idx[:, 1]).tolist()[:20], 'cols', np.unique(idx[:, 2]).tolist()[:20])
        print(' a', a[tuple(idx[0])], 'b', b[tuple(idx[0])])
    # a constant-per-plane input tells which plane is read where
    xc = np.zeros_like(x); xc[:, :, 0] = 1.0
    os.environ['TSM_STEM_PLANAR'] = '1'
