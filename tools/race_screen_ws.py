"""Race screen for the weight-stationary kernels (conv3x3_ws_kernel<false|true>, conv3x3_ws128_kernel, conv1x1_ws[n]_kernel: hand-placed counted
vmcnt waits, LDS-DMA into buffers that are re-used, one barrier per tile, a wave-private residual ring): every case is
run REPS times back to back and every result is compared bit for bit with the igemm path.  A schedule that reads a
staged patch / ring slot too early passes most runs and fails some; this looks for the some.
  per-op cases: the 'ws' tile against the 64x64 tile through tsm_conv_bn_act;
  engine cases: TSM_FUSE_CONV23=1 (bf16: layer1.1 / layer1.2 run conv3x3_ws_kernel<true>) against TSM_FUSE_CONV23=0,
  block outputs and logits, several forwards per engine; TSM_FUSE_BLOCK=1 (bneck_ws_kernel) against TSM_FUSE_BLOCK=0."""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from workoutdetector_amd.engine import TsmEngine, conv_bn_act_nhwc  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
# (channels, frames, h, w): tiles per workgroup from < 1 to several; ragged tiles; frames smaller than a tile
# (stride 2: conv3x3_ws128_kernel<true>, layer2.0's conv2 -- one M-tile pair per tile, the accumulator sets alternate)
for ch, n, h, w, stride in [(64, 64, 64, 64, 1), (64, 200, 64, 64, 1), (64, 96, 56, 56, 1), (64, 37, 23, 18, 1), (64, 700, 8, 8, 1),
                            (128, 64, 32, 32, 1), (128, 300, 32, 32, 1), (128, 96, 28, 28, 1), (128, 37, 23, 18, 1), (128, 700, 8, 8, 1),
                            (128, 64, 64, 64, 2), (128, 112, 64, 64, 2), (128, 96, 56, 56, 2), (128, 37, 23, 18, 2), (128, 700, 8, 8, 2),
                            (128, 11, 33, 200, 2)]:
    g = torch.Generator().manual_seed(n + ch)
    x = torch.randn(n, h, w, ch, generator=g).cuda()
    wt = (torch.randn(ch, ch, 3, 3, generator=g) * (2.0 / (9 * ch)) ** 0.5).cuda()
    bn = [torch.rand(ch, generator=g).cuda() + 0.5, torch.randn(ch, generator=g).cuda() * 0.1,
          torch.randn(ch, generator=g).cuda() * 0.1, torch.rand(ch, generator=g).cuda() + 0.5]
    os.environ['TSM_CONV_TILE'] = '64x64'
    ref = conv_bn_act_nhwc(x, wt, *bn, stride=stride, relu=True, dtype='bf16')
    os.environ['TSM_CONV_TILE'] = 'ws'
    fails = 0
    for _ in range(REPS):
        got = conv_bn_act_nhwc(x, wt, *bn, stride=stride, relu=True, dtype='bf16')
        fails += int(not torch.equal(got, ref))
    bad += fails
    print(f'per-op  ch={ch} n={n} {h}x{w} stride {stride}: {fails}/{REPS} runs differ from the 64x64 tile')
# conv1x1_ws_kernel / conv1x1_wsn_kernel (conv1 of layer1 / layer2 / layer3.0 with the fused temporal shift)
for cin, cout, n, h, w, T in [(256, 64, 64, 64, 64, 16), (256, 64, 96, 56, 56, 8), (64, 64, 64, 64, 64, 16), (256, 64, 27, 7, 5, 3),
                               (64, 64, 700, 8, 8, 7), (256, 128, 64, 64, 64, 16), (512, 128, 96, 32, 32, 16), (512, 256, 96, 32, 32, 8),
                               (512, 128, 27, 7, 5, 3)]:
    g = torch.Generator().manual_seed(n + cin + cout + 1)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).cuda()
    bn = [torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda() * 0.1,
          torch.randn(cout, generator=g).cuda() * 0.1, torch.rand(cout, generator=g).cuda() + 0.5]
    os.environ['TSM_CONV_TILE'] = '64x64'
    ref = conv_bn_act_nhwc(x, wt, *bn, stride=1, relu=True, shift_segments=T, dtype='bf16')
    os.environ['TSM_CONV_TILE'] = 'ws'
    fails = 0
    for _ in range(REPS):
        got = conv_bn_act_nhwc(x, wt, *bn, stride=1, relu=True, shift_segments=T, dtype='bf16')
        fails += int(not torch.equal(got, ref))
    bad += fails
    print(f'per-op  1x1 {cin}->{cout} n={n} {h}x{w} T={T}: {fails}/{REPS} runs differ from the 64x64 tile')
os.environ.pop('TSM_CONV_TILE', None)
sd = make_state_dict(0, 12)
for b, t, s in [(4, 8, 224), (8, 16, 256), (3, 8, 96), (2, 8, 90)]:
    x = np.random.default_rng(b + s).standard_normal((b, t, 3, s, s)).astype(np.float32)
    outs = {}
    for flag in ('0', '1'):
        os.environ['TSM_FUSE_CONV23'] = flag
        eng = TsmEngine(num_segments=t, height=s, width=s, max_clips=b, state_dict=sd, dtype='bf16')
        outs[flag] = []
        for _ in range(1 if flag == '0' else REPS):
            outs[flag].append((eng.forward_tap(x, 'layer1.1'), eng.forward_tap(x, 'layer1.2'), eng.run(None, {'input': x})[0]))
        eng.close()
    fails = sum(int(not all(np.array_equal(a, r) for a, r in zip(o, outs['0'][0]))) for o in outs['1'])
    bad += fails
    print(f'engine  bf16 B={b} T={t} {s}x{s}: {fails}/{REPS} fused forwards differ from the separate launches')
os.environ.pop('TSM_FUSE_CONV23', None)
# bneck_ws_kernel (whole layer1.1 / layer1.2 block: wave-private LDS-DMA slots re-armed a step ahead, a four-row line buffer
# and a mid tile re-used every step behind two barriers, counted vmcnt over [residual | input | stores]): TSM_FUSE_BLOCK=1
# against the separate launches, many frames per workgroup (several engines' worth of steps), ragged and tiny frames
for b, t, s in [(8, 16, 256), (24, 16, 256), (32, 8, 224), (3, 8, 96), (2, 8, 90), (40, 3, 64)]:
    x = np.random.default_rng(7 * b + s).standard_normal((b, t, 3, s, s)).astype(np.float32)
    outs = {}
    for flag in ('0', '1'):
        os.environ['TSM_FUSE_BLOCK'] = flag
        eng = TsmEngine(num_segments=t, height=s, width=s, max_clips=b, state_dict=sd, dtype='bf16')
        outs[flag] = []
        for _ in range(1 if flag == '0' else max(4, REPS // 3)):
            outs[flag].append((eng.forward_tap(x, 'layer1.1'), eng.forward_tap(x, 'layer1.2'), eng.run(None, {'input': x})[0]))
        eng.close()
    fails = sum(int(not all(np.array_equal(a, r) for a, r in zip(o, outs['0'][0]))) for o in outs['1'])
    bad += fails
    print(f'engine  bf16 B={b} T={t} {s}x{s}: {fails}/{len(outs["1"])} whole-block forwards differ from the separate launches')
os.environ.pop('TSM_FUSE_BLOCK', None)
print('RACE SCREEN', 'FAILED' if bad else 'clean')
sys.exit(1 if bad else 0)
