"""Measure the BASELINE.json configurations that are not bench.py's headline, on ONE MI355X.

  config 3  pull_up-shaped stream: 1080 frames of 360x206 uint8 -> 135 clips, end to end
            (H2D of the uint8 frames, fused HIP transform, engine, softmax/threshold, pred_to_count),
            plus the streaming variant: latency of one 8-frame window at batch 1
  config 4  RepCount-val-shaped workload: 100 videos, >= 10 021 clips in total, end to end per video
            (the 8-GPU sharding of this config is covered by tests/test_distributed_cpu.py; here: 1 GPU)
  config 5  T=16, 256x256, 64 clips per GPU, TSM_DTYPE_BF16 (and the exact-f32 mode on the same shape)

    python tools/bench_configs.py [--dtype f32|bf16x3] > gpurun_out/configs.json

  config 4, strong scaling over N GPUs (fixed job: the 100 RepCount-val videos / 10 062 clips of the committed
  annotation, `inference_dataset(shard='global')`: videos to ranks longest-first, no collective in the loop, one
  exchange at the end, rank 0 writes the 100 JSON files):

    python tools/bench_configs.py --config 4 --gpus N [--dtype ...]      (starts its own N ranks, like bench.py)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == '__main__' and '--gpus' in sys.argv and 'WORLD_SIZE' not in os.environ:
    _n = int(sys.argv[sys.argv.index('--gpus') + 1])
    if _n > 1:      # start the ranks BEFORE this process imports torch or touches the GPU (same helper as bench.py)
        import bench
        sys.exit(bench.self_launch(_n, script=__file__))

import numpy as np
import torch

from workoutdetector_amd import inference_count as ic  # noqa: E402
from workoutdetector_amd.counting import pred_to_count, scores_to_preds  # noqa: E402
from workoutdetector_amd.engine import TsmEngine  # noqa: E402
from workoutdetector_amd.transform import build_test_transform  # noqa: E402
from workoutdetector_amd.weights import make_state_dict  # noqa: E402


def sync():
    torch.cuda.synchronize()


def config3(eng, reps=5):
    rng = np.random.default_rng(0)
    vid = torch.from_numpy(rng.integers(0, 256, size=(1080, 360, 206, 3), dtype=np.uint8))
    tf = build_test_transform(False)
    ic.video_clip_logits(eng, vid, tf)          # warm-up (tile autotune for 32- and 7-clip batches)
    sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        logits = ic.video_clip_logits(eng, vid, tf, batch_clips=32)
        states = scores_to_preds(logits.tolist())
        count, _ = pred_to_count(states, 8)
        ts.append(time.perf_counter() - t0)
    n = logits.shape[0]
    # streaming: one non-overlapping 8-frame window at a time (batch 1), frames already on the host
    win = vid[:8]
    lat = []
    for i in range(12):
        t0 = time.perf_counter()
        ic.inference_video(eng, win.float(), transform=tf)
        lat.append(time.perf_counter() - t0)
    lat = sorted(lat[2:])
    # the engine-native streaming path: StreamBatcher (fused HIP transform, windows of all live streams in one batch)
    from workoutdetector_amd.streaming import StreamBatcher
    frames = vid.numpy()
    sb = StreamBatcher(eng, max_batch=32)
    lat1 = []
    for w in range(14):                                   # ONE stream: latency from the 8th frame to its state
        for f in frames[w * 8:w * 8 + 8]:
            sb.push('solo', f)
        t0 = time.perf_counter()
        sb.step()
        lat1.append(time.perf_counter() - t0)
    lat1 = sorted(lat1[2:])
    sb = StreamBatcher(eng, max_batch=32)
    for s_ in range(32):                                  # 32 live streams, 4 windows each per step
        for f in frames[s_ * 32:s_ * 32 + 32]:
            sb.push(s_, f)
    sync()
    t0 = time.perf_counter()
    out = sb.step()
    t32 = time.perf_counter() - t0
    n_win = sum(len(v) for v in out.values())
    return {'clips': int(n), 'end_to_end_s_median': float(np.median(ts)), 'clips_per_s': n / float(np.median(ts)),
            'count': int(count), 'stream_window_latency_ms_median': 1e3 * lat[len(lat) // 2],
            'stream_batcher_one_stream_window_ms_median': 1e3 * lat1[len(lat1) // 2],
            'stream_batcher_32_streams_windows_per_s': n_win / t32,
            'note': 'end to end = uint8 frames on the host -> H2D -> tsm_preprocess -> tsm_forward (batches of 32) -> '
                    'D2H logits -> softmax/threshold -> pred_to_count; window latency = reference-style '
                    'inference_video (torch transform + host round trip) at batch 1; stream_batcher_* = '
                    'workoutdetector_amd.streaming.StreamBatcher: host uint8 frames -> H2D -> tsm_preprocess -> engine -> states'}


def config4(eng, total_clips=10021, n_videos=100):
    rng = np.random.default_rng(0)
    # clip counts per video: lognormal-ish spread with the annotated lower bound as the total
    w = rng.lognormal(0.0, 0.6, n_videos)
    clips = np.maximum(8, np.round(w / w.sum() * total_clips)).astype(int)
    clips[-1] += max(0, total_clips - int(clips.sum()))
    tf = build_test_transform(False)
    gen = torch.Generator().manual_seed(0)
    # warm-up: the tile autotuner runs once per power-of-two bucket of the batch size (ragged last batches hit all of them)
    eng.warmup()
    ic.video_clip_logits(eng, torch.randint(0, 256, (64, 360, 206, 3), dtype=torch.uint8, generator=gen), tf)
    sync()
    # synthetic "decoded" videos are made up front (no decoder offline); in batches so host memory stays bounded
    n, t_total, counts = 0, 0.0, []
    for g0 in range(0, n_videos, 20):
        vids = []
        for c in clips[g0:g0 + 20]:
            frames = int(c) * 8 - int(rng.integers(0, 8))
            vids.append(torch.randint(0, 256, (frames, 360, 206, 3), dtype=torch.uint8, generator=gen))
        sync()
        t0 = time.perf_counter()
        for _, st in ic.prefetch_staged(eng, enumerate(vids)):
            logits = ic.staged_clip_logits(eng, st, tf, batch_clips=32)
            counts.append(pred_to_count(scores_to_preds(logits.tolist()), 8)[0])
            n += logits.shape[0]
        t_total += time.perf_counter() - t0
    return {'videos': n_videos, 'clips': int(n), 'end_to_end_s': t_total, 'clips_per_s': n / t_total,
            'note': 'per video: pin + H2D of the uint8 even frames (prefetched one video ahead on a side stream), fused '
                    'transform, engine in batches of 32 (ragged last batch), one D2H, counter; synthetic frame '
                    'generation is outside the timed region (no decoder offline)'}


def config5(dtype, steps=10, warmup=3, batch=64):
    eng = TsmEngine(num_segments=16, height=256, width=256, max_clips=batch, state_dict=make_state_dict(0, 12), dtype=dtype)
    x = torch.randn(batch, 16, 3, 256, 256, device='cuda')
    out = torch.empty(batch, 12, device='cuda')
    for _ in range(warmup):
        eng.forward_device(x, out=out)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward_device(x, out=out)
    sync()
    dt = (time.perf_counter() - t0) / steps
    eng.close()
    gflop = 170.826
    return {'dtype': dtype, 'clips_per_gpu': batch, 'ms_per_step': 1e3 * dt, 'clips_per_s': batch / dt,
            'algorithmic_tflops': gflop * batch / dt / 1e3}


def config4_scaling(args):
    """One JSON line: clips/s of the whole fixed job (strong scaling), max over ranks of the wall time from the first
    video read to rank 0's last JSON file, plus the modelled plan efficiency and the per-rank clip loads."""
    import shutil
    import tempfile
    import pandas as pd
    import torch.distributed as dist
    from workoutdetector_amd import distributed as tdist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    rehearsal = os.environ.get('TSM_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    anno = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'repcount_annotation.csv'), index_col=0)
    val = anno[(anno.split == 'val') & anno.class_.isin(['situp', 'push_up', 'pull_up', 'jump_jack', 'squat', 'front_raise'])]
    if args.videos:
        val = val.head(args.videos)
    frames = {}
    for _, r in val.iterrows():
        reps = [int(v) for v in str(r['reps']).split()] if int(r['count']) > 0 else []
        frames[r['name']] = max(max(reps) if reps else 0, 16)
    fh, fw = (int(v) for v in args.frame_size.split('x'))
    # No decoder / dataset offline: every "decoded" video is a window of one seeded pool of random uint8 frames, so the
    # reader costs a view, while pinning, H2D, the transform and everything behind them move real bytes.
    pool = torch.randint(0, 256, (max(frames.values()) + 8, fh, fw, 3), dtype=torch.uint8,
                         generator=torch.Generator().manual_seed(0))

    def reader(path):
        return pool[:frames[os.path.basename(path)]]

    def counter(path):
        return frames[os.path.basename(path)]

    base = os.environ.get('TSM_BENCH_TMP') or tempfile.gettempdir()
    root = os.path.join(base, 'tsm_config4_job_%s' % os.environ.get('MASTER_PORT', 'solo'))
    if rank == 0:
        shutil.rmtree(root, ignore_errors=True)
        os.makedirs(root)
        val.to_csv(os.path.join(root, 'annotation.csv'))
    if world > 1:
        dist.barrier()
        if rank != 0:
            dist.barrier()       # rank 0 tunes first; the others read its choices (shared TSM_TUNE_CACHE)
    eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12), dtype=args.dtype, device=local_rank)
    if not args.cold:
        eng.warmup()             # (--cold: the job itself tunes / reads the tune cache while its first frames are staged)
    if world > 1 and rank == 0:
        dist.barrier()
    first = {}
    real_forward = eng.forward_device

    def timed_forward(*a, **k):          # host time of the job's first forward launch (cold-job head)
        first.setdefault('t', time.perf_counter())
        return real_forward(*a, **k)

    eng.forward_device = timed_forward
    import contextlib
    import io
    sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        ic.inference_dataset(eng, ['val'], os.path.join(root, 'out'), checkpoint='seed0', data_root=root,
                             video_reader=reader, frame_counter=counter, batch_clips=32, shard='global')
    sync()
    dt = torch.tensor([time.perf_counter() - t0], device='cuda')
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    clips = [len(range(0, f, 8)) for f in frames.values()]
    owner = tdist.plan_video_shards(clips, world)
    if rank == 0:
        n_files = len(os.listdir(os.path.join(root, 'out')))
        assert n_files == len(frames), (n_files, len(frames))
        load = [sum(c for c, o in zip(clips, owner) if o == r) for r in range(world)]
        print(json.dumps({
            'metric': 'clips/sec, RepCount-val-shaped dataset run end to end (BASELINE.json configs[3])',
            'value': None if rehearsal else round(sum(clips) / float(dt.item()), 2), 'unit': 'clips/s', 'n_gpus': world,
            'scaling': 'strong', 'dtype': args.dtype, 'videos': len(frames), 'clips': sum(clips),
            'job_s': round(float(dt.item()), 4), 'frame_size': args.frame_size, 'clips_per_rank': load,
            'cold': bool(args.cold), 'tune_cache': os.environ.get('TSM_TUNE_CACHE', 'default (~/.cache/tsm_hip/tune_cache.txt)'),
            'first_forward_after_s': round(first['t'] - t0, 4) if 't' in first else None,
            'plan_efficiency': round(tdist.shard_efficiency(clips, owner, world), 4),
            'lockstep_round_robin_efficiency': round(tdist.lockstep_efficiency(clips, world), 4),
            **({'rehearsal': True} if rehearsal else {}),
            'note': 'inference_dataset(shard="global"): whole videos to ranks longest-first, per rank pin + H2D of uint8 '
                    'frames (prefetched two videos ahead), fused HIP transform, engine in full cross-video batches of 32, '
                    'no collective inside the loop, every rank writes the JSON files of its own videos under the GPU work '
                    'of the videos behind them (asynchronous D2H per video), ONE exchange at the end (video table + padded '
                    'logits); time = max over ranks, barrier to last file'}), flush=True)
        shutil.rmtree(root, ignore_errors=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--config', type=int, default=0, help='4: the strong-scaling dataset run (with --gpus N)')
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--videos', type=int, default=0, help='config 4: only the first N val videos (rehearsals)')
    ap.add_argument('--frame-size', default='360x206', help='config 4: HxW of the synthetic decoded frames')
    ap.add_argument('--cold', action='store_true', help='config 4: no warmup() before the job (a cold process: the job tunes '
                                                        'or reads the tune cache itself, under the staging of its first frames)')
    args = ap.parse_args()
    if args.config == 4:
        config4_scaling(args)
        return
    eng = TsmEngine(max_clips=32, state_dict=make_state_dict(0, 12), dtype=args.dtype)
    res = {'engine_dtype': args.dtype, 'config3': config3(eng), 'config4': config4(eng)}
    eng.close()
    res['config5_bf16'] = config5('bf16')
    res['config5_f32'] = config5('f32', steps=4, warmup=2)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
