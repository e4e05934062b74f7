"""First contact with RCCL on the box's GPU (VERDICT r1 'what's missing' #1).

The multi-GPU exchange of this path is one all-gather of per-clip logits over ``torch.distributed``'s ``nccl`` backend
(= RCCL on ROCm).  A GPU test box has ONE GPU, so these tests bring up ONE-rank nccl groups in child processes and force
the collective branch (``distributed.set_force_collective`` / ``TSM_BENCH_FORCE_COLLECTIVE=1``): RCCL loads, the
``device_id=`` initialisation works, device tensors go through ``all_gather_into_tensor``, pad / trim is right, and
``bench.py``'s N > 1 step structure (barrier, collective inside the step, max over ranks) runs end to end.  What one GPU
cannot show -- xGMI transport between ranks -- stays with the driver's 8-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'TSM_BENCH_REHEARSAL', 'TSM_BENCH_FORCE_COLLECTIVE'):
        env.pop(k, None)
    env.update(extra)
    return env


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith('{')]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_one_rank_nccl_group_drives_the_collective_branch(hip_lib, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_rccl_child.py'), str(tmp_path), str(_free_port())],
                         capture_output=True, text=True, env=_env(), timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rep = _last_json(out.stdout)
    assert rep['ok'] and rep['backend'] == 'nccl' and rep['world'] == 1 and rep['collective_calls'] == 14


def test_bench_step_through_a_one_rank_rccl_group(hip_lib):
    """bench.py's multi-GPU step (forward + all-gather inside the timed region, barriers, MAX all-reduce of the time,
    exchange latency) on a 1-rank nccl group: everything `torchrun --nproc-per-node N bench.py --gpus N` executes
    except a second rank."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                          '--batch', '4', '--no-alt', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900,
                         env=_env(TSM_BENCH_FORCE_COLLECTIVE='1', MASTER_PORT=str(_free_port())))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line['n_gpus'] == 1 and line['value'] > 0 and 'rehearsal' not in line
    assert 'backend nccl = RCCL' in line['exchange']['collective'] and line['exchange']['avg_us'] > 0
    assert '1-rank RCCL' in line['config']['parallelism']
    assert line['step_ms']['min'] <= line['step_ms']['median'] <= line['step_ms']['max']


def test_rehearsal_line_is_labelled_and_carries_no_value(hip_lib):
    """TSM_BENCH_REHEARSAL=1 (two ranks sharing cuda:0 over gloo) only rehearses the control flow: the line must say
    so and must not offer a throughput value or call the exchange RCCL."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--batch', '2', '--no-alt', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(TSM_BENCH_REHEARSAL='1'), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = _last_json(out.stdout)
    assert line['rehearsal'] is True and line['value'] is None and line['n_gpus'] == 2
    assert line['config']['parallelism'].startswith('REHEARSAL') and 'NOT RCCL' in line['exchange']['collective']
    assert 'RCCL (nccl' not in line['config']['parallelism']


def test_bench_starts_its_own_ranks_when_no_launcher_is_given(hip_lib):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset (the driver's plain command form): the parent spawns the two
    ranks before touching the GPU, they share ONE tune cache (rank 0 tunes, rank 1 reads), rank 0's single JSON line
    comes through the parent's stdout and carries per-rank step medians.  Rehearsal: both ranks on this box's one GPU."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                          '--batch', '2', '--no-alt', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900,
                         env=_env(TSM_BENCH_REHEARSAL='1'), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    json_lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(json_lines) == 1, out.stdout[-3000:]
    line = json.loads(json_lines[0])
    assert line['rehearsal'] is True and line['n_gpus'] == 2 and line['value'] is None
    ex = line['exchange']
    assert len(ex['per_rank_step_ms_median']) == 2 and all(v > 0 for v in ex['per_rank_step_ms_median'])
    assert len(ex['per_rank_forward_kernel_ms_median']) == 2 and 'rank 0 tunes' in ex['tune_cache']
