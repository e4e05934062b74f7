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
    ic.inference_dataset(StubModel(), ['test'], single, checkpoint='stub', data_root=str(root))
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
    ic.inference_dataset(StubModel(), ['test'], single, checkpoint='stub', data_root=str(root))
    _spawn(_dataset_by_videos_worker, str(root), sharded, world=world)
    assert sorted(os.listdir(single)) == sorted(os.listdir(sharded)) and len(os.listdir(single)) == 3
    for f in os.listdir(single):
        assert json.load(open(os.path.join(single, f))) == json.load(open(os.path.join(sharded, f)))
    # and the single-process form of the same mode
    alone = str(tmp_path / 'alone')
    ic.inference_dataset(StubModel(), ['test'], alone, checkpoint='stub', data_root=str(root), shard='videos')
    for f in os.listdir(single):
        assert json.load(open(os.path.join(single, f))) == json.load(open(os.path.join(alone, f)))
