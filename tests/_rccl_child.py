"""Child process of tests/test_rccl_gpu.py: ONE rank, backend nccl (= RCCL on ROCm), on the box's GPU.

A one-rank group makes every collective the identity, so the product normally skips it; with
``distributed.set_force_collective(True)`` the real branch runs: ``init_process_group('nccl', device_id=...)``,
device-tensor ``all_gather_into_tensor``, pad / trim of ragged blocks, the on-GPU meta + logits gathers of the
video-sharded dataset path.  Prints one JSON line; exits non-zero on any mismatch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(tmp, port):
    import numpy as np
    import pandas as pd
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    report = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
    from tests._stub import synthetic_video
    from workoutdetector_amd import distributed as tdist
    from workoutdetector_amd import inference_count as ic
    from workoutdetector_amd.engine import TsmEngine
    from workoutdetector_amd.weights import make_state_dict

    # 1. the raw collective on device tensors
    x = torch.arange(7 * 12, dtype=torch.float32, device='cuda').reshape(7, 12)
    out = torch.empty_like(x)
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    assert torch.equal(out, x)
    # 2. the product's gathers: skipped without the switch, executed (and exact) with it
    assert not tdist.collective_enabled()
    tdist.set_force_collective(True)
    assert tdist.collective_enabled() and tdist.on_rccl()
    calls = {'n': 0}
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls['n'] += 1
        assert a[0].is_cuda and a[1].is_cuda, 'the nccl path must hand device tensors to the collective'
        return real(*a, **k)

    dist.all_gather_into_tensor = counting
    g = tdist.gather_clip_logits(x, 7)
    assert g.is_cuda and torch.equal(g, x) and calls['n'] == 1
    assert torch.equal(tdist.all_gather_logits(x[:3]), x[:3]) and calls['n'] == 2

    # 3. inference_dataset through the collective branch, both sharding forms, against the plain single-process run
    anno = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'repcount_annotation.csv'), index_col=0)
    rows = anno[anno['name'].isin(['stu1_40.mp4', 'stu5_32.mp4', 'stu3_53.mp4'])].copy()
    rows['name'] = [n.replace('.mp4', '.npy') for n in rows['name']]
    root = os.path.join(tmp, 'RepCount')
    os.makedirs(os.path.join(root, 'videos', 'test'))
    rows.to_csv(os.path.join(root, 'annotation.csv'))
    for i, name in enumerate(rows['name']):
        np.save(os.path.join(root, 'videos', 'test', name), synthetic_video(60 + i, (90, 17, 41)[i], 96, 64, period=20))
    eng = TsmEngine(num_class=12, max_clips=8, state_dict=make_state_dict(0, 12))
    tdist.set_force_collective(False)
    plain = os.path.join(tmp, 'plain')
    ic.inference_dataset(eng, ['test'], plain, checkpoint='seed0', data_root=root, batch_clips=8)
    assert calls['n'] == 2
    tdist.set_force_collective(True)
    for shard, expect_calls in (('clips', 3), ('videos', 6), ('global', 3)):     # one gather per video / meta + logits per round / plan checksum + video table + logits per JOB
        before = calls['n']
        out_dir = os.path.join(tmp, shard)
        ic.inference_dataset(eng, ['test'], out_dir, checkpoint='seed0', data_root=root, batch_clips=8, shard=shard)
        assert calls['n'] - before == expect_calls, (shard, calls['n'] - before)
        assert sorted(os.listdir(out_dir)) == sorted(os.listdir(plain)) and len(os.listdir(plain)) == 3
        for f in os.listdir(plain):
            assert json.load(open(os.path.join(plain, f))) == json.load(open(os.path.join(out_dir, f))), (shard, f)
    report['collective_calls'] = calls['n']
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    report['ok'] = True
    print(json.dumps(report), flush=True)


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))
