"""Race screen for conv31_fused_kernel (conv3 + residual of block b and shift + conv1 of block b + 1 as one launch: counted
vmcnt waits over LDS-DMA'd weight chunks that are re-filled every chunk, a staged / register-loaded t2 tile, a residual
stream one or two chunks ahead, three barriers per chunk, an LDS tile whose rows other waves read shifted by a frame):
every case runs REPS forwards back to back on ONE engine with TSM_FUSE_C3C1=1 and every result -- the block outputs behind
each kind of site, the next block's conv1 tap and the logits -- is compared bit for bit with an engine that may not fuse.
A wait that is one operation short passes most runs and fails some; this looks for the some, on many tiles per workgroup
(config-5 batches), ragged last tiles (224 x 224: 784 pixels per frame), T = 2 .. 32 and tiny frames.
    python tools/race_screen_c31.py [REPS]"""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
STAGES = ('layer2.1', 'layer2.2.conv1', 'layer2.3', 'layer3.0.conv1', 'layer3.1', 'layer3.2.conv1', 'layer3.5')
sd = make_state_dict(0, 12)
bad = 0
for b, t, s in [(16, 16, 256), (32, 16, 256), (32, 8, 224), (9, 8, 224), (6, 4, 96), (5, 2, 64), (3, 32, 128), (7, 8, 90)]:
    x = np.random.default_rng(11 * b + s + t).standard_normal((b, t, 3, s, s)).astype(np.float32)
    outs = {}
    for flag in ('0', '1'):
        os.environ['TSM_FUSE_C3C1'] = flag
        eng = TsmEngine(num_segments=t, height=s, width=s, max_clips=b, state_dict=sd, dtype='bf16')
        outs[flag] = []
        for rep in range(1 if flag == '0' else REPS):
            taps = tuple(eng.forward_tap(x, st) for st in STAGES) if (flag == '0' or rep % 5 == 0) else ()
            outs[flag].append(taps + (eng.run(None, {'input': x})[0],))
        if flag == '1':     # the forced switch sets no tile-code bit: count the conv1 launches the forward did NOT make
            eng.set_layer_timing(1)
            eng.run(None, {'input': x})
            lt = eng.layer_times_ms(0)
            sites = [k for k, v in lt.items() if k.endswith('.conv1') and not k.startswith('layer1.') and v < 0]
        eng.close()
    ref = outs['0'][0]
    fails = 0
    for o in outs['1']:
        want = ref if len(o) == len(ref) else ref[-1:]
        fails += int(not all(np.array_equal(a, r) for a, r in zip(o, want)))
    bad += fails
    print(f'engine  bf16 B={b} T={t} {s}x{s}: {fails}/{REPS} fused forwards differ from the separate launches '
          f'({len(sites)} cross-block launches per forward)', flush=True)
os.environ.pop('TSM_FUSE_C3C1', None)
print('RACE SCREEN', 'FAILED' if bad else 'clean')
sys.exit(1 if bad else 0)
