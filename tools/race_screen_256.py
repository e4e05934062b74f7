"""Race screen for conv_bf16_256_kernel and its persistent form conv_bf16_256p_kernel (hand-placed vmcnt / barrier
schedule, staggered wave groups; the persistent form adds tile boundaries inside the schedule): every shape is run REPS
times back to back on the tile and every result is compared bit for bit with the 128 x 128 tile's.  A schedule that reads
a staged half-operand too early passes most runs and fails some; this looks for the some.

    python tools/race_screen_256.py [reps [256x256|256x256p]]"""
import os
import sys

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from workoutdetector_amd.engine import conv_bn_act_nhwc  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
TILE = sys.argv[2] if len(sys.argv) > 2 else '256x256'
shapes = [(64, 16, 16, 256, 256, 3, 1, 0), (256, 8, 8, 512, 512, 3, 1, 0), (128, 16, 16, 1024, 256, 1, 1, 16),
          (32, 16, 16, 256, 256, 3, 2, 0), (16, 20, 20, 64, 256, 3, 1, 0), (64, 8, 8, 2048, 512, 1, 1, 16),
          (37, 7, 9, 256, 256, 3, 1, 0),
          # more tiles than workgroups (the persistent form's tile boundaries): 1x1 with two K-tiles, 3x3, shifted conv1
          (416, 16, 16, 128, 512, 1, 1, 0), (600, 16, 16, 64, 256, 3, 1, 0), (1024, 16, 16, 1024, 256, 1, 1, 16),
          (512, 16, 16, 256, 256, 3, 1, 0),
          # conv3 + residual (T = -1 marks "with a residual"): the persistent form's sub-slab epilogue and its prefetched residual
          (416, 16, 16, 128, 512, 1, 1, -1), (300, 16, 16, 256, 1024, 1, 1, -1), (128, 8, 8, 512, 2048, 1, 1, -1),
          (37, 9, 7, 128, 512, 1, 1, -1)]
bad = 0
for n, h, w, cin, cout, k, stride, T in shapes:
    with_res, T = T < 0, max(T, 0)
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).cuda()
    bn = [torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda() * 0.1,
          torch.randn(cout, generator=g).cuda() * 0.1, torch.rand(cout, generator=g).cuda() + 0.5]
    res = torch.randn(n, (h - 1) // stride + 1, (w - 1) // stride + 1, cout, generator=g).cuda() if with_res else None
    os.environ['TSM_CONV_TILE'] = '128x128'
    ref = conv_bn_act_nhwc(x, wt, *bn, stride=stride, relu=True, residual=res, shift_segments=T, dtype='bf16')
    os.environ['TSM_CONV_TILE'] = TILE
    tag = ' +res' if with_res else ''
    fails = 0
    for _ in range(REPS):
        got = conv_bn_act_nhwc(x, wt, *bn, stride=stride, relu=True, residual=res, shift_segments=T, dtype='bf16')
        fails += int(not torch.equal(got, ref))
    bad += fails
    print(f'n={n} {h}x{w} cin={cin} cout={cout} k={k} s={stride} T={T}{tag}: {fails}/{REPS} runs of {TILE} differ from the 128x128 tile')
print('RACE SCREEN', 'FAILED' if bad else 'clean')
sys.exit(1 if bad else 0)
