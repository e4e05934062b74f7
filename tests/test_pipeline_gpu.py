"""Config 3 of BASELINE.json: a pull_up-shaped frame stream through the whole path on the GPU --
clip windows -> transform -> HIP engine -> softmax/threshold -> pred_to_count -- against the CPU oracle.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import counting_oracle, transform_oracle, tsm_oracle
from tests._stub import synthetic_video
from tests._util import assert_close, fit_probe_fc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def probe_engine(hip_lib, sd0):
    """Seed-0 trunk + a classifier fitted (on the oracle) to flip between classes 4/5 with brightness."""
    from workoutdetector_amd.engine import TsmEngine
    sd = dict(sd0)
    probe = torch.from_numpy(synthetic_video(11, 96, 90, 52, period=24))
    sd['fc.weight'], sd['fc.bias'] = fit_probe_fc(sd0, probe)
    eng = TsmEngine(num_class=12, num_segments=8, max_clips=32, state_dict=sd)
    yield eng, sd
    eng.close()


def test_stream_counts_identical_to_oracle(probe_engine):
    """Non-trivial integer repetition count, identical between the HIP path and the oracle path; logits within
    fp32 rtol 1e-3.  Stream is portrait like RepCount's stu* clips (scaled down to keep the CPU oracle short)."""
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.transform import build_test_transform
    eng, sd = probe_engine
    vid = torch.from_numpy(synthetic_video(23, 140, 90, 52, period=24))      # 18 clips, tail zero-padded
    got = ic.video_clip_logits(eng, vid, build_test_transform(False), batch_clips=32)
    want = torch.cat([tsm_oracle.tsm_forward(sd, transform_oracle.clip_to_input(transform_oracle.make_clip(vid, s)))
                      for s in range(0, 140, 8)])
    assert_close(got.numpy(), want.numpy(), rtol=1e-3, atol_scale=1e-5, what='stream logits')
    from workoutdetector_amd.counting import pred_to_count, scores_to_preds
    states = scores_to_preds(got.tolist())
    assert states == counting_oracle.scores_to_preds(want.tolist())
    count, reps = pred_to_count(states, 8)
    assert (count, reps) == counting_oracle.pred_to_count(states, 8)
    assert count >= 3 and set(states) <= {4, 5, -1}, (states, count)         # ~140/24 periods of bright->dark


def test_full_size_stream_properties(probe_engine, tmp_path):
    """BASELINE config 3 at full size: 1080 frames of 360x206 uint8 -> 135 clips.  Size-independent
    properties instead of the oracle: batch invariance (B=32 chunks vs one clip at a time), the
    reference's clip-by-clip inference_video path, streaming == offline counts, JSON round trip."""
    from workoutdetector_amd import eval as tsm_eval
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.counting import pred_to_count, scores_to_preds
    from workoutdetector_amd.transform import build_test_transform
    eng, _ = probe_engine
    vid = torch.from_numpy(synthetic_video(5, 1080, 360, 206, period=36))
    tf = build_test_transform(False)
    batched = ic.video_clip_logits(eng, vid, tf, batch_clips=32)
    assert tuple(batched.shape) == (135, 12)
    single = ic.video_clip_logits(eng, vid, tf, batch_clips=1)
    assert torch.equal(batched, single)
    for s in (0, 8 * 67, 8 * 134):                                            # incl. the zero-padded tail clip
        ref_style = ic.inference_video(eng, ic.make_clip(vid, s).cuda(), transform=tf)
        # reference-style path = torch transform + NCHW hand-over; batched path = fused HIP transform (K8):
        # same math, different fp32 rounding of the bilinear weights
        np.testing.assert_allclose(np.float32([v for _, v in ref_style]), batched[s // 8].numpy(), rtol=1e-5, atol=1e-4)
    states = scores_to_preds(batched.tolist())
    count, reps = pred_to_count(states, 8)
    assert count >= 20                                                        # 1080 / 36 = 30 brightness periods
    scores = ic.scores_dict(batched, 1080)
    path = tmp_path / 'v.score.json'
    json.dump(dict(scores=scores, action='pull_up'), open(path, 'w'))
    back = json.load(open(path))['scores']
    assert tsm_eval.preds_from_scores(back, softmax=True) == states
    # streaming windows (stride 8, no overlap) see different frames than the sparse offline windows, but must
    # be self-consistent with their own offline evaluation
    c2, r2 = ic.count_by_video_model(eng, iter(vid[:400].numpy()))
    st2 = []
    for i in range(0, 400, 8):
        sc = [v for _, v in ic.inference_video(eng, vid[i:i + 8].float(), transform=tf)]
        st2.append(scores_to_preds([sc])[0])
    assert (c2, r2) == pred_to_count(st2, 8)


def test_inference_dataset_on_gpu_engine(probe_engine, tmp_path, golden_dir):
    """inference_dataset counterpart end to end with the real engine on two synthetic RepCount videos."""
    import pandas as pd
    from workoutdetector_amd import inference_count as ic
    eng, sd = probe_engine
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[anno['name'].isin(['stu1_40.mp4', 'stu5_32.mp4'])].copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    frames = {'stu1_40.npy': 336, 'stu5_32.npy': 52}
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(40 + i, frames[name], 120, 68, period=28))
    out = str(tmp_path / 'out')
    ic.inference_dataset(eng, ['test'], out, checkpoint='seed0+probe', data_root=str(root))
    d = json.load(open(os.path.join(out, 'stu1_40.npy.score.json')))
    assert d['total_frames'] == 336 and len(d['scores']) == 42 and d['action'] == 'pull_up'
    # spot-check three clips against the oracle
    vid = torch.from_numpy(np.load(root / 'videos' / 'test' / 'stu1_40.npy'))
    for s in (0, 160, 328):
        want = tsm_oracle.tsm_forward(sd, transform_oracle.clip_to_input(transform_oracle.make_clip(vid, s)))[0]
        got = np.float32([d['scores'][str(s)][str(c)] for c in range(12)])
        assert_close(got, want.numpy(), rtol=1e-3, atol_scale=1e-5, what=f'clip {s}')


def _play_plain(StreamBatcher, eng, vids, max_batch, every):
    sb = StreamBatcher(eng, max_batch=max_batch)
    ev = {k: [] for k in vids}
    for t in range(240):
        for k, v in vids.items():
            if t < len(v):
                sb.push(k, v[t])
        if t % every == every - 1:
            for k, e in sb.step().items():
                ev[k] += e
    for k, e in sb.step().items():
        ev[k] += e
    return ev, {k: sb.result(k) for k in vids}


def test_stream_batcher_on_gpu_engine(probe_engine):
    """several live streams of different frame sizes share one engine; windows of all
    streams go through the fused HIP transform + one tsm_forward per step.  Exact: batched == one window at a
    time (batch invariance); the per-stream counts equal an offline pred_to_count of that stream's states."""
    from workoutdetector_amd.counting import pred_to_count
    from workoutdetector_amd.streaming import StreamBatcher
    eng, _ = probe_engine
    vids = {'a': synthetic_video(31, 240, 90, 52, period=24), 'b': synthetic_video(32, 200, 120, 68, period=32),
            'c': synthetic_video(33, 168, 90, 52, period=20)}

    def play(max_batch, every, **kw):
        sb = StreamBatcher(eng, max_batch=max_batch, **kw)
        ev = {k: [] for k in vids}
        for t in range(240):
            for k, v in vids.items():
                if t < len(v):
                    sb.push(k, v[t])
            if t % every == every - 1:
                for k, e in sb.step().items():
                    ev[k] += e
        for k, e in sb.step().items():
            ev[k] += e
        assert sb.pinned_bytes <= sb.max_pinned_bytes
        for k in list(vids):
            sb.close(k)
        assert sb.pinned_bytes == 0 and not sb._free          # closing the last stream of a size frees its idle buffers
        return ev, {k: sb.streams.get(k) for k in vids}, sb

    def play_res(*a, **kw):
        results = {}
        ev, _, sb = play(*a, **kw, on_window=lambda sid, w, state, count: results.__setitem__(sid, count))
        return ev, results

    # a pinned budget of ~2.5 windows and 1 idle buffer per size: the producer runs 5 windows ahead of step(), so most
    # windows fall back to pageable memory (bounded page-locked memory) -- same events, same counts
    ev_cap, res_cap = play_res(32, 40, max_pinned_bytes=300_000, max_free_per_shape=1)
    ev32, res32 = play_res(32, 40)
    assert ev_cap == ev32 and res_cap == res32
    # open / close cycles at a resolution another open stream still uses must not grow the idle list past the cap
    sb = StreamBatcher(eng, max_batch=4, max_free_per_shape=2)
    sb.push('keep', vids['a'][0])
    win = 8 * 90 * 52 * 3
    for i in range(6):
        for t in range(11):                       # one complete window that never runs + a half-filled one
            sb.push(('x', i), vids['a'][t])
        sb.close(('x', i))
        idle = sb._free.get((90, 52, 3), [])
        assert len(idle) <= 2 and sb.pinned_bytes == (1 + len(idle)) * win, (i, len(idle), sb.pinned_bytes)
    sb.close('keep')
    assert sb.pinned_bytes == 0 and not sb._free
    ev32, res32 = _play_plain(StreamBatcher, eng, vids, 32, 40)          # 5 windows x 3 streams per step -> batches of up to 15 mixed-size windows
    ev1, res1 = _play_plain(StreamBatcher, eng, vids, 1, 8)
    assert ev32 == ev1 and res32 == res1
    for k, v in vids.items():
        states = [s for _, s, _ in ev32[k]]
        assert len(states) == len(v) // 8
        assert res32[k] == pred_to_count(states, 8) == counting_oracle.pred_to_count(states, 8)
    assert res32['a'][0] >= 5 and res32['b'][0] >= 3


def _gpu_dataset_worker(rank, world, port, root, out_dir, shard):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)       # one GPU on the test box: gloo, both ranks on cuda:0
    try:
        from workoutdetector_amd import inference_count as ic
        from workoutdetector_amd.engine import TsmEngine
        from workoutdetector_amd.weights import make_state_dict
        eng = TsmEngine(num_class=12, max_clips=8, state_dict=make_state_dict(0, 12))
        ic.inference_dataset(eng, ['test'], out_dir, checkpoint='seed0', data_root=root, batch_clips=8, shard=shard)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('shard', ['clips', 'videos'])
def test_two_rank_dataset_inference_with_real_engines(hip_lib, tmp_path, golden_dir, shard):
    """The N > 1 dataset path with REAL engines: two processes (gloo; both on the box's single GPU), clips or whole
    videos sharded, logits all-gathered, rank 0 writes -- JSON identical to the single-process run, bit for bit
    (batch invariance makes the per-rank batch shapes irrelevant)."""
    import socket

    import pandas as pd
    import torch.multiprocessing as mp
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.engine import TsmEngine
    from workoutdetector_amd.weights import make_state_dict
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[anno['name'].isin(['stu1_40.mp4', 'stu5_32.mp4', 'stu3_53.mp4'])].copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(60 + i, (90, 17, 41)[i], 96, 64, period=20))
    single, sharded = str(tmp_path / 'single'), str(tmp_path / 'sharded')
    eng = TsmEngine(num_class=12, max_clips=8, state_dict=make_state_dict(0, 12))
    ic.inference_dataset(eng, ['test'], single, checkpoint='seed0', data_root=str(root), batch_clips=8, shard='clips')
    eng.close()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_dataset_worker, args=(2, port, str(root), sharded, shard), nprocs=2, join=True)
    assert sorted(os.listdir(single)) == sorted(os.listdir(sharded)) and len(os.listdir(single)) == 3
    for f in os.listdir(single):
        assert json.load(open(os.path.join(single, f))) == json.load(open(os.path.join(sharded, f))), f


def test_oversized_videos_are_staged_in_pieces(probe_engine, monkeypatch):
    """A video whose even frames exceed inference_count.MAX_STAGE_BYTES (1 GiB by default: long 1080p videos) is not
    pinned / uploaded whole: the clip range is walked in pieces, each re-staging the frames its windows overlap.
    Same logits bit for bit, including the zero-padded tail clip and a piece boundary inside the video."""
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.transform import build_test_transform
    eng, _ = probe_engine
    vid = torch.from_numpy(synthetic_video(77, 333, 90, 52, period=24))      # 42 clips, 167 even frames of 14 040 B
    tf = build_test_transform(False)
    whole = ic.video_clip_logits(eng, vid, tf, batch_clips=16)
    monkeypatch.setattr(ic, 'MAX_STAGE_BYTES', 40 * 90 * 52 * 3)             # room for 40 even frames -> 8 clips per piece
    pieces = ic.video_clip_logits(eng, vid, tf, batch_clips=16)
    assert tuple(whole.shape) == (42, 12) and torch.equal(whole, pieces)
    part = ic.video_clip_logits(eng, vid, tf, clip_range=(5, 31), batch_clips=16)   # a rank's block of a sharded video
    assert torch.equal(part, whole[5:31])


@pytest.mark.gpu
def test_dataset_loop_pieces_shrink_to_the_staging_bound(probe_engine, tmp_path, golden_dir, monkeypatch):
    """The dataset loop (shard='global') stages videos in pieces; a piece shrinks until it fits MAX_STAGE_BYTES.  With a
    bound of 12 even frames a piece is ONE clip (8 even frames + the pad frame): score files byte-identical to the
    video-at-a-time path with whole-video staging, i.e. piece boundaries (every clip is one), the overlap of
    neighbouring pieces' frames and the cross-video batcher change no bit."""
    import pandas as pd
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.repcount import CLASSES
    eng, _ = probe_engine
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[(anno['split'] == 'test') & anno['class_'].isin(CLASSES)].head(3).copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    for i, (name, frames) in enumerate(zip(rows['name'], (77, 130, 9))):
        np.save(root / 'videos' / 'test' / name, synthetic_video(20 + i, frames, 90, 52, period=24))
    whole, pieces = str(tmp_path / 'whole'), str(tmp_path / 'pieces')
    ic.inference_dataset(eng, ['test'], whole, checkpoint='seed0', data_root=str(root), batch_clips=8, shard='clips')
    monkeypatch.setattr(ic, 'MAX_STAGE_BYTES', 12 * 90 * 52 * 3)
    got = ic.inference_dataset(eng, ['test'], pieces, checkpoint='seed0', data_root=str(root), batch_clips=8)
    files = sorted(os.listdir(whole))
    assert files == sorted(os.listdir(pieces)) and len(files) == 3
    for f in files:
        assert open(os.path.join(whole, f)).read() == open(os.path.join(pieces, f)).read(), f
    assert sorted(int(v.shape[0]) for v in got.values()) == [2, 10, 17]
