"""The N > 1 path on CPU: world_size-2 gloo process groups (one process per rank, like one per GPU)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from workoutdetector_amd import distributed as tdist


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(rank, world, port, fn, args):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def _spawn(fn, *args, world=2):
    mp.spawn(_run, args=(world, _free_port(), fn, args), nprocs=world, join=True)


def test_shard_ranges_cover_everything_once():
    for n in [0, 1, 2, 7, 8, 9, 135, 10021]:
        for world in [1, 2, 3, 8]:
            blocks = [tdist.shard_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert all(hi - lo <= tdist.per_rank(n, world) for lo, hi in blocks)
            assert tdist.shard_list(list(range(n)), world, world - 1) == list(range(*blocks[-1]))


def _gather_worker(rank, world, tmp):
    for n in [1, 2, 5, 16, 17]:
        full = torch.arange(n * 12, dtype=torch.float32).reshape(n, 12) * 0.5
        lo, hi = tdist.shard_range(n, world, rank)
        got = tdist.gather_clip_logits(full[lo:hi].clone(), n)
        assert torch.equal(got, full), (n, rank)
    per = 4
    local = torch.full((per, 12), float(rank))
    out = tdist.all_gather_logits(local)
    assert tuple(out.shape) == (world * per, 12)
    assert torch.equal(out[:per], torch.zeros(per, 12)) and torch.equal(out[per:2 * per], torch.ones(per, 12))
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('1')


def test_all_gather_of_ragged_clip_logits(tmp_path):
    _spawn(_gather_worker, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ['ok0', 'ok1']


def _dataset_worker(rank, world, root, out_dir):
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic
    ic.inference_dataset(StubModel(), ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=3, shard='clips')


def test_sharded_inference_dataset_equals_single_process(tmp_path, golden_dir):
    """Clips of each video split over 2 ranks + one all-gather == the single-process result, bit for bit;
    only rank 0 writes."""
    import pandas as pd
    from tests._stub import StubModel, synthetic_video
    from workoutdetector_amd import inference_count as ic
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[anno['name'].isin(['stu1_40.mp4', 'stu5_32.mp4'])].copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    frames = {'stu1_40.npy': 77, 'stu5_32.npy': 9}                         # 10 clips / 2 clips (ragged over 2 ranks)
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(i, frames[name], 40, 30))
    single, sharded = str(tmp_path / 'single'), str(tmp_path / 'sharded')
    ic.inference_dataset(StubModel(), ['test'], single, checkpoint='stub', data_root=str(root), shard='clips')
    _spawn(_dataset_worker, str(root), sharded)
    assert sorted(os.listdir(single)) == sorted(os.listdir(sharded))
    for f in os.listdir(single):
        assert json.load(open(os.path.join(single, f))) == json.load(open(os.path.join(sharded, f)))


def _dataset_by_videos_worker(rank, world, root, out_dir):
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic
    ic.inference_dataset(StubModel(), ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=4, shard='videos')


@pytest.mark.parametrize('world', [2, 3])
def test_video_sharded_inference_dataset_equals_single_process(tmp_path, golden_dir, world):
    """shard='videos': whole videos round-robin over the ranks (3 videos over 2 ranks -> a half-empty last round;
    over 3 ranks -> one round), counts + padded logits all-gathered, rank 0 writes: identical JSON files."""
    import pandas as pd
    from tests._stub import StubModel, synthetic_video
    from workoutdetector_amd import inference_count as ic
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[anno['name'].isin(['stu1_40.mp4', 'stu5_32.mp4', 'stu3_53.mp4'])].copy()
    assert len(rows) == 3
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(i, (77, 9, 41)[i], 40 + 2 * i, 30))   # sizes differ
    single, sharded = str(tmp_path / 'single'), str(tmp_path / 'sharded')
    ic.inference_dataset(StubModel(), ['test'], single, checkpoint='stub', data_root=str(root), shard='clips')
    _spawn(_dataset_by_videos_worker, str(root), sharded, world=world)
    assert sorted(os.listdir(single)) == sorted(os.listdir(sharded)) and len(os.listdir(single)) == 3
    for f in os.listdir(single):
        assert json.load(open(os.path.join(single, f))) == json.load(open(os.path.join(sharded, f)))
    # and the single-process form of the same mode
    alone = str(tmp_path / 'alone')
    ic.inference_dataset(StubModel(), ['test'], alone, checkpoint='stub', data_root=str(root), shard='videos')
    for f in os.listdir(single):
        assert json.load(open(os.path.join(single, f))) == json.load(open(os.path.join(alone, f)))


def _dataset_global_worker(rank, world, root, out_dir, batch):
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic
    calls = {'n': 0}
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)

    dist.all_gather_into_tensor = counting
    model = StubModel()
    ic.inference_dataset(model, ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=batch)   # default shard
    # no collective inside the loop: the plan checksum up front, then the whole job exchanges twice (video table + status
    # words, then the logits), whatever its size
    assert calls['n'] == 3, calls
    open(os.path.join(out_dir, f'calls{rank}'), 'w').write(str(model.calls))


GLOBAL_FRAMES = (77, 9, 41, 160, 8, 23, 95, 64, 130, 17)      # 10 / 2 / 6 / 20 / 1 / 3 / 12 / 8 / 17 / 3 clips


@pytest.mark.parametrize('world', [2, 3, 8])
def test_globally_sharded_inference_dataset_equals_single_process(tmp_path, golden_dir, world):
    """shard='global' (the default): whole videos to ranks longest-first by clip count, cross-video batches, every rank
    writes the files of its own videos as they finish, ONE exchange at the end: the JSON files are byte-identical to the
    single-process video-at-a-time run (shard='clips') at W = 2, 3 and 8 (W = 8 > the number of long videos: some ranks
    own one short video), and so are those of the single-process run of the same mode."""
    import pandas as pd
    from tests._stub import StubModel, synthetic_video
    from workoutdetector_amd import inference_count as ic
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    from workoutdetector_amd.repcount import CLASSES
    rows = anno[(anno['split'] == 'test') & anno['class_'].isin(CLASSES)].head(len(GLOBAL_FRAMES)).copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(i, GLOBAL_FRAMES[i], 40 + 2 * (i % 3), 30))
    single, sharded = str(tmp_path / 'single'), str(tmp_path / 'sharded')
    ic.inference_dataset(StubModel(), ['test'], single, checkpoint='stub', data_root=str(root), shard='clips')
    os.makedirs(sharded)
    _spawn(_dataset_global_worker, str(root), sharded, 4, world=world)
    files = sorted(f for f in os.listdir(sharded) if f.endswith('.json'))
    assert files == sorted(os.listdir(single)) and len(files) == len(GLOBAL_FRAMES)
    for f in files:
        assert open(os.path.join(single, f)).read() == open(os.path.join(sharded, f)).read(), f
    # full cross-video batches: the ranks together ran ceil(clips_on_rank / 4) forwards each, not one ragged tail per video
    clips = [len(range(0, f, 8)) for f in GLOBAL_FRAMES]
    owner = tdist.plan_video_shards(clips, world)
    for r in range(world):
        mine = sum(c for c, o in zip(clips, owner) if o == r)
        assert int(open(os.path.join(sharded, f'calls{r}')).read()) == -(-mine // 4), (r, mine)
    # the single-process form of the same mode
    alone = str(tmp_path / 'alone')
    ic.inference_dataset(StubModel(), ['test'], alone, checkpoint='stub', data_root=str(root), shard='global', batch_clips=5)
    for f in files:
        assert open(os.path.join(single, f)).read() == open(os.path.join(alone, f)).read(), f


def _dataset_global_failing_worker(rank, world, root, out_dir):
    """Rank 1's model raises on its second forward: EVERY rank must come out of inference_dataset with an exception (rank 1
    with its own, the others with one naming rank 1) instead of blocking in the final all-gather."""
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic

    class Flaky(StubModel):
        def run(self, output_names, feed):
            if rank == 1 and self.calls >= 1:
                raise OSError('decoder died')
            return super().run(output_names, feed)

    try:
        ic.inference_dataset(Flaky(), ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=4)
    except OSError as exc:
        assert rank == 1 and 'decoder died' in str(exc)
        open(os.path.join(out_dir, f'raised{rank}'), 'w').write('own')
    except RuntimeError as exc:
        assert rank != 1 and 'rank(s) [1] failed' in str(exc), str(exc)
        open(os.path.join(out_dir, f'raised{rank}'), 'w').write('peer')


def _dataset_global_plan_failing_worker(rank, world, root, out_dir):
    """Rank 2's frame counter raises while the plan is laid out (before the FIRST exchange): every rank must come out of
    inference_dataset with an exception instead of blocking in the plan-checksum all-gather."""
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic

    def counter(path):
        if rank == 2:
            raise OSError('no index for ' + os.path.basename(path))
        return int(np.load(path, mmap_mode='r').shape[0])

    try:
        ic.inference_dataset(StubModel(), ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=4, frame_counter=counter)
    except OSError as exc:
        assert rank == 2 and 'no index' in str(exc)
        open(os.path.join(out_dir, f'raised{rank}'), 'w').write('own')
    except RuntimeError as exc:
        assert rank != 2 and 'rank(s) [2] failed while planning' in str(exc), str(exc)
        open(os.path.join(out_dir, f'raised{rank}'), 'w').write('peer')


def _dataset_global_warmup_failing_worker(rank, world, root, out_dir):
    """Rank 0's model fails in warmup() (a tsm_tune HIP error / OOM in the real engine) -- after the plan exchange, before the
    loop: it must ride the status row of the final exchange like a failure inside the loop."""
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic

    class ColdFail(StubModel):
        def warmup(self, sizes):
            if rank == 0:
                raise MemoryError('tune scratch')
            return self

    # (the warm-up only runs for a model on a device: give the stub one the way _engine_device reads it)
    orig = ic._engine_device
    ic._engine_device = lambda m: None
    try:
        model = ColdFail()
        run_global = ic._run_global

        def run_with_warm(model_, items, mine, counts, make_pieces, warm, *rest):
            return run_global(model_, items, mine, counts, make_pieces, (lambda: model_.warmup([4])), *rest)
        ic._run_global = run_with_warm
        ic.inference_dataset(model, ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=4)
    except MemoryError:
        assert rank == 0
        open(os.path.join(out_dir, f'raised{rank}'), 'w').write('own')
    except RuntimeError as exc:
        assert rank != 0 and 'rank(s) [0] failed' in str(exc), str(exc)
        open(os.path.join(out_dir, f'raised{rank}'), 'w').write('peer')
    finally:
        ic._engine_device = orig


def test_global_sharding_failure_on_one_rank_reaches_every_rank(tmp_path, golden_dir):
    """with no collective inside the loop, a rank that raises must still enter the final exchange (error marker
    in the video table) -- the other ranks then raise too instead of waiting for ever; and ranks whose shard plans differ
    fail before any work."""
    import pandas as pd
    from tests._stub import synthetic_video
    from workoutdetector_amd.repcount import CLASSES
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[(anno['split'] == 'test') & anno['class_'].isin(CLASSES)].head(6).copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(i, GLOBAL_FRAMES[i], 40, 30))
    out_dir = str(tmp_path / 'out')
    os.makedirs(out_dir)
    _spawn(_dataset_global_failing_worker, str(root), out_dir, world=3)
    assert sorted(f for f in os.listdir(out_dir) if f.startswith('raised')) == ['raised0', 'raised1', 'raised2']
    assert open(os.path.join(out_dir, 'raised1')).read() == 'own' and open(os.path.join(out_dir, 'raised0')).read() == 'peer'
    # ... a failure BEFORE the first exchange (planning) and one between the exchanges but outside the loop (warm-up)
    for worker, owner_rank in ((_dataset_global_plan_failing_worker, 2), (_dataset_global_warmup_failing_worker, 0)):
        out2 = str(tmp_path / worker.__name__)
        os.makedirs(out2)
        _spawn(worker, str(root), out2, world=3)
        assert sorted(f for f in os.listdir(out2) if f.startswith('raised')) == ['raised0', 'raised1', 'raised2'], worker.__name__
        for r in range(3):
            assert open(os.path.join(out2, f'raised{r}')).read() == ('own' if r == owner_rank else 'peer'), (worker.__name__, r)


def _dataset_global_anon_worker(rank, world, root, out_dir):
    from tests._stub import StubModel
    from workoutdetector_amd import inference_count as ic

    class Anonymous(StubModel):          # a duck-typed session that does not say how many classes it has
        num_class = None

    got = ic.inference_dataset(Anonymous(), ['test'], out_dir, checkpoint='stub', data_root=root, batch_clips=4)
    assert sorted(got) == sorted(f[:-len('.score.json')] for f in os.listdir(out_dir) if f.endswith('.json'))
    assert all(tuple(v.shape)[1] == 12 for v in got.values())


def test_global_sharding_with_a_rank_that_owns_no_video(tmp_path, golden_dir):
    """Two videos over three ranks: rank 2 runs nothing and -- with a model that does not expose num_class -- cannot
    know the class count; it must still take part in both exchanges (the count rides in the video table), every rank
    gets the whole job's logits back, and the files equal the single-process run."""
    import pandas as pd
    from tests._stub import StubModel, synthetic_video
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.repcount import CLASSES
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    rows = anno[(anno['split'] == 'test') & anno['class_'].isin(CLASSES)].head(2).copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = tmp_path / 'RepCount'
    (root / 'videos' / 'test').mkdir(parents=True)
    rows.to_csv(root / 'annotation.csv')
    for i, name in enumerate(rows['name']):
        np.save(root / 'videos' / 'test' / name, synthetic_video(i, (50, 19)[i], 40, 30))
    single, sharded = str(tmp_path / 'single'), str(tmp_path / 'sharded')
    ic.inference_dataset(StubModel(), ['test'], single, checkpoint='stub', data_root=str(root), shard='clips')
    _spawn(_dataset_global_anon_worker, str(root), sharded, world=3)
    for f in os.listdir(single):
        assert open(os.path.join(single, f)).read() == open(os.path.join(sharded, f)).read(), f


def test_global_shard_plan_is_balanced_on_the_repcount_val_distribution(golden_dir):
    """BASELINE config 4's own distribution (100 val videos, 2-327 clips, 10 062 in all, from the committed annotation):
    the longest-first plan keeps every rank within 5 % of the mean up to W = 8 with no exchange until the end, where
    round-2's lock-stepped round-robin (an exchange per round of W videos) modelled at 0.72 / 0.56 / 0.45."""
    import pandas as pd
    anno = pd.read_csv(f'{golden_dir}/repcount_annotation.csv', index_col=0)
    val = anno[(anno.split == 'val') & anno.class_.isin(['situp', 'push_up', 'pull_up', 'jump_jack', 'squat', 'front_raise'])]
    clips = []
    for _, r in val.iterrows():
        reps = [int(v) for v in str(r['reps']).split()] if int(r['count']) > 0 else []
        clips.append(len(range(0, max(max(reps) if reps else 0, 16), 8)))
    assert len(clips) == 100 and sum(clips) == 10062
    for world, old in ((2, 0.73), (4, 0.57), (8, 0.45)):
        owner = tdist.plan_video_shards(clips, world)
        assert sorted(set(owner)) == list(range(world))
        assert tdist.shard_efficiency(clips, owner, world) >= 0.95
        assert tdist.lockstep_efficiency(clips, world) <= old
    assert tdist.plan_video_shards(clips, 1) == [0] * 100
    # estimates from the annotation alone (no frame files): the same plan on every rank, any order of evaluation
    assert tdist.plan_video_shards(clips, 8) == tdist.plan_video_shards(list(clips), 8)


def test_bench_self_launch_propagates_a_failing_rank_without_hanging():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own ranks (bench.self_launch).  Without a GPU every
    rank stops at the product's "no CPU fallback" assertion: the parent must come back with a non-zero code (never
    hang on the ranks, never print a JSON line, never fall back to a CPU measurement)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip('this is the no-GPU failure path')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode != 0
    assert 'needs a GPU' in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
