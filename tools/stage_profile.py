"""Where the per-video time of the dataset path goes (config 4): staging on the host vs GPU compute.

    python tools/stage_profile.py [--dtype bf16x3]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from workoutdetector_amd import inference_count as ic  # noqa: E402
from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.transform import build_test_transform  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='bf16x3')
    ap.add_argument('--frames', type=int, default=800)
    a = ap.parse_args()
    eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12), dtype=a.dtype)
    tf = build_test_transform(False)
    vid = torch.randint(0, 256, (a.frames, 360, 206, 3), dtype=torch.uint8)
    ic.video_clip_logits(eng, vid, tf)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        st = ic.stage_video(eng, vid)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        logits = ic.staged_clip_logits(eng, st, tf, batch_clips=32)
        t3 = time.perf_counter()
        print(f'{a.dtype} clips={logits.shape[0]} stage(host)={1e3 * (t1 - t0):.1f} ms  h2d wait={1e3 * (t2 - t1):.1f} ms  '
              f'transform+engine+d2h={1e3 * (t3 - t2):.1f} ms  -> {logits.shape[0] / (t3 - t2):.0f} clips/s compute-only')
    # pieces of the compute leg
    from workoutdetector_amd.engine import preprocess_frames
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr = preprocess_frames(st.frames, resize=tf.size, crop=tf.crop, scale_255=tf.scale_255, layout=eng.packed_layout)
    torch.cuda.synchronize()
    print(f'preprocess {st.frames.shape[0]} frames: {1e3 * (time.perf_counter() - t0):.2f} ms')
    eng.close()


if __name__ == '__main__':
    main()
