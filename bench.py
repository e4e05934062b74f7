#!/usr/bin/env python3
"""Throughput bench for the TSM-R50 clip-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (HIP engine forward, tsm_forward through the C ABI) over one
batch of 32 synthetic clips [32, 8, 3, 224, 224] fp32 that is already resident in HBM (BASELINE.json
configs[1]).  N > 1: one process per GPU -- either under a launcher (`python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment)
or, when WORLD_SIZE is unset, started by this script itself before it touches the GPU (`self_launch`);
clips are independent units so every rank runs its own batch (weak scaling) and the only exchange is the
RCCL all-gather of per-clip logits, which is inside the timed step.  All ranks share one tile-tuning pass
(TSM_TUNE_CACHE: rank 0 tunes, the others read).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel is the kernel the engine's autotuner runs the 3x3 convolutions of
                layer2..layer4 on (13 launches per forward, one third of the forward's time; `kernel_of` maps the
                tuner's tile code to the kernel name rocprofv3 prints; if the tuner split them over several
                kernels, the one with the larger total).  Every plain 3x3 launch does the same algorithmic work,
                115.6 MMAC/frame x 2 x frames (59.19 GFLOP at batch 32); a launch that also runs its block's conv3
                is a different kernel and is priced with both convs; `achieved` = work / the launches' average duration, measured with HIP-event
                pairs recorded around each launch on the launch stream during the timed steps
                (tsm_set_layer_timing / tsm_layer_times), against the exact-fp32 MFMA peak (157.3 TFLOP/s,
                MI355X_MICROARCH.md).  `forward_achieved` / `forward_frac` price the whole forward
                (65.395 GFLOP/clip, SURVEY.md section 8d, all 53 conv launches + pool/head kernels).
  cpu_baseline  the CPU oracle (oracle/tsm_oracle.py, torch-CPU fp32, the same graph; the reference's
                own onnxruntime CPU path cannot run here) timed on this box's host cores, batch 1 like
                the reference (utils/inference_count.py:272), bounded to ~15 s; median (`value`) and best, with the CPU
                model, the load average and torch's thread count beside them (a shared host: the figure moves with them)
  config5       BASELINE.json configs[4] behind the headline (default command only): TSM_DTYPE_BF16, T = 16, 256 x 256, 64 clips
                per GPU through the same call, with its own tuning, timed steps, `roofline` (dominant kernel against the
                dense bf16 MFMA peak, forward_frac, PMC traffic, forward HBM rate) and `parity` (bf16-storage oracle); the
                headline fields stay those of configs[1].  --no-config5 skips it (profiling runs).
  build_id      tsm_build_id() of the library that ran == the sha of the source tree it was built from (the loader refuses
                any other); `tune`: where the tile choices came from
  parity        the logits the TIMED steps produced (the output of the last timed step; what
                utils/inference_count.py:273-275 would hand to the counter), checked on rank 0 AFTER the timed region
                against the CPU oracle on a few of the timed clips: max |err| / logit scale against the bar of the parity
                tests (fp32 / split-bf16: rtol 1e-3 + 1e-5 of the scale, the fp32 oracle; bf16: 1e-2 of the scale and
                the same arg-max, the bf16-storage oracle).  A failed check makes the run fail, not just the line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_CLIP_T8_224 = 65.395          # 2 * (4.0871 GMAC/frame + 24576) * 8, SURVEY.md section 8(d)
PEAK_F32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA", dense
PEAK_HBM_GBS = 8000.0                   # MI355X_MICROARCH.md, HBM3E ~8 TB/s


def flops_per_clip(t, h, w, num_class=12):
    from workoutdetector_amd.flops import flops_per_clip as f   # algorithmic work, SURVEY section 8d
    return f(t, h, w, num_class)


def host_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the box's total."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get('TSM_BENCH_CPU_THREADS', '16')))


def measured_traffic(b, t, h, w, kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json), only when
    they were collected on this exact configuration AND on this exact kernel source (entries are stamped with the
    sha of csrc/ they were measured on); bench.py cannot run PMC passes itself.  Returns (entry, stale_note)."""
    from workoutdetector_amd.build import csrc_sha16
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except (OSError, ValueError):
        return None, None
    sha, stale = csrc_sha16(), None
    for e in d.get('entries', []):
        c = e.get('config', {})
        if ((c.get('clips_per_gpu'), c.get('num_segments'), c.get('height'), c.get('width')) == (b, t, h, w)
                and e.get('kernel') == kernel):
            if e.get('csrc_sha16') == sha:
                return e, None
            stale = (f"profiles/traffic.json holds {e.get('hbm_bytes_per_launch')} B/launch for this kernel measured on "
                     f"csrc {e.get('csrc_sha16', 'unstamped (round 1)')}; current csrc is {sha}: re-run tools/pmc_traffic.sh")
    return None, stale


def cpu_baseline(sd_np, t, h, w, budget_s=15.0):
    import torch
    from oracle import tsm_oracle
    from workoutdetector_amd.weights import to_torch
    cores = host_cores()
    torch.set_num_threads(cores)
    load0 = os.getloadavg()
    sd = to_torch(sd_np)
    x = torch.randn(1, t, 3, h, w, generator=torch.Generator().manual_seed(0))
    tsm_oracle.tsm_forward(sd, x, n_segment=t)  # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end or len(times) < 3:
        t0 = time.perf_counter()
        tsm_oracle.tsm_forward(sd, x, n_segment=t)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    # SURVEY 8d also asks for batch 8 (what a batched CPU deployment of the same graph would get): a short second sample
    x8 = torch.randn(8, t, 3, h, w, generator=torch.Generator().manual_seed(1))
    tsm_oracle.tsm_forward(sd, x8, n_segment=t)
    times8 = []
    t_end = time.perf_counter() + budget_s / 2
    while time.perf_counter() < t_end or len(times8) < 3:
        t0 = time.perf_counter()
        tsm_oracle.tsm_forward(sd, x8, n_segment=t)
        times8.append(time.perf_counter() - t0)
    times8.sort()
    cpu_model = None
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.lower().startswith('model name'):
                cpu_model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round(1.0 / med, 3), 'unit': 'clips/s', 'cores': cores, 'kind': 'port',
            'best': round(1.0 / times[0], 3), 'worst': round(1.0 / times[-1], 3),
            'batch8_value': round(8.0 / times8[len(times8) // 2], 3), 'batch8_best': round(8.0 / times8[0], 3),
            'cpu_model': cpu_model, 'host_cpus': os.cpu_count(), 'torch_threads': torch.get_num_threads(),
            'loadavg_before_after': [round(v, 2) for v in load0[:2]] + [round(v, 2) for v in os.getloadavg()[:2]],
            'note': 'a shared host: the share of the cores this process really gets moves with the load average '
                    '(1-min, 5-min before and after the sample); `best` is the least disturbed run',
            'sample': f'{len(times)} x 1 clip [1,{t},3,{h},{w}] fp32, batch 1 like the reference, median (`value`); '
                      f'{len(times8)} x 8 clips, median (`batch8_value`); torch-CPU oracle of the same graph '
                      f'(reference onnxruntime CPU path not runnable here)'}


PARITY_CLIPS = 3     # clips of the timed batch held against the oracle (first, middle, last)


def oracle_logits(sd_np, clips_cpu, t, dtypes):
    """{'f32': fp32-oracle logits, 'bf16': bf16-storage-oracle logits} of a few of the timed clips (test infrastructure
    used as the CHECKER of the timed output, after the timed region, on rank 0 only)."""
    import torch
    from oracle import tsm_oracle
    from workoutdetector_amd.weights import to_torch
    torch.set_num_threads(host_cores())
    sd = to_torch(sd_np)
    want = {}
    if any(d in ('f32', 'bf16x3') for d in dtypes):
        want['f32'] = tsm_oracle.tsm_forward(sd, clips_cpu, n_segment=t).numpy()
    if 'bf16' in dtypes:
        want['bf16'] = tsm_oracle.tsm_forward_bf16(sd, clips_cpu, n_segment=t).numpy()
    return want


def parity_of(got, want, dtype):
    """The bars of the parity tests (tests/_util.py): fp32 / split-bf16 |err| <= 1e-3 |want| + 1e-5 scale against the
    fp32 oracle; bf16 |err| <= 1e-2 scale and the same arg-max against the bf16-storage oracle."""
    import numpy as np
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(want['bf16' if dtype == 'bf16' else 'f32'], dtype=np.float64)
    scale = float(np.abs(ref).max())
    err = np.abs(got - ref)
    if dtype == 'bf16':
        ok = bool(np.isfinite(got).all() and (err <= 1e-2 * scale).all() and (got.argmax(1) == ref.argmax(1)).all())
        bar, against = 1e-2, 'bf16-storage oracle (oracle.tsm_oracle.tsm_forward_bf16), bar = 1e-2 of the logit scale + same arg-max'
    else:
        ok = bool(np.isfinite(got).all() and (err <= 1e-3 * np.abs(ref) + 1e-5 * scale).all())
        bar, against = 1e-3, 'fp32 oracle (oracle.tsm_oracle.tsm_forward), bar = rtol 1e-3 + 1e-5 of the logit scale'
    return {'clips_checked': int(got.shape[0]), 'max_err_over_scale': float('%.3g' % (err.max() / scale)), 'bar': bar,
            'ok': ok, 'against': against, 'logit_scale': float('%.4g' % scale),
            'what': 'logits written by the last TIMED step, compared after the timed region (rank 0)'}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n, script=None):
    """`python bench.py --gpus N` with N > 1 and no launcher: this process -- which has not imported torch, let alone
    touched the GPU -- starts N fresh children of itself, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run would set them; the reference's launcher for comparison: tools/dist_train.sh:3-10), lets rank
    0's JSON line through on the inherited stdout and exits with the worst child code.  All ranks share one
    TSM_TUNE_CACHE file: rank 0 tunes, the others read its choices (main(), `tuned_engine`)."""
    import subprocess
    import tempfile
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    tune = os.environ.get('TSM_TUNE_CACHE')
    tmpdir = None
    if not tune:
        tmpdir = tempfile.mkdtemp(prefix='tsm_bench_')
        tune = os.path.join(tmpdir, 'tune_cache.txt')
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, TSM_TUNE_CACHE=tune,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + sys.argv[1:], env=env))
    worst, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and deadline is None:
                deadline = time.monotonic() + 30.0       # a rank died: the others are stuck in a collective
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                              # exactly the PIDs started above
        time.sleep(0.2)
    for p in procs:
        rc = p.wait()
        worst = rc if (rc != 0 and worst == 0) else worst
    if tmpdir:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)
    return worst


def kernel_of(tile, dtype, cmid, stride=1):
    """rocprofv3's name (minus `void tsm::` and the parameter list) of the kernel a 3x3 launch with tile name `tile`
    (TsmEngine.tile_name) runs, and whether that launch also does the block's conv3 (+ residual)."""
    fused = tile.endswith('+conv3')
    main = tile.replace('+conv3', '').split('/')[0]
    if main == 'ws':
        if cmid == 64:
            return 'conv3x3_ws_kernel<%s>' % ('true' if fused else 'false'), fused
        return 'conv3x3_ws128_kernel<%s>' % ('true' if stride == 2 else 'false'), fused   # <S2>
    if fused:
        return 'conv23_fused_kernel<%d, %s>' % (cmid, 'true' if dtype == 'bf16x3' else 'false'), True
    if main == '256x256':
        return 'conv_bf16_256_kernel<3, false, false, false>', False
    if main == '256x256p':
        return 'conv_bf16_256p_kernel<3, false, false, false>', False
    waves = '1, 1' if main == '32x32' else '4, 2' if main.endswith('w8') else '2, 2'
    prec_id = {'f32': 0, 'bf16x3': 1, 'bf16': 2}[dtype]
    # template arguments: BM, BN, WGM, WGN, KS, SHIFT, RES, PREC, DUAL, SEG (fp32 long-K layers accumulate K in segments)
    return 'conv_igemm<%s, %s, 3, false, false, %d, false, %s>' % (
        main.replace('w8', '').replace('x', ', '), waves, prec_id, 'true' if dtype == 'f32' else 'false'), False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU per step')
    ap.add_argument('--segments', type=int, default=8)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle check of the timed logits (profiling runs)')
    ap.add_argument('--no-alt', action='store_true', help='skip the second run in the other precision mode')
    ap.add_argument('--no-config5', action='store_true',
                    help='skip the BASELINE configs[4] leg (bf16, T=16, 256x256, 64 clips per GPU) behind the headline')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16x3', 'bf16'],
                    help='f32: exact-fp32 MFMA; bf16x3: split-bf16 storage, 3 bf16 MFMAs per product')
    ap.add_argument('--config', type=int, default=2, choices=[2, 5],
                    help='BASELINE.json configuration of the HEADLINE fields: 2 = the defaults (configs[1], what the metric is '
                         'quoted on; a configs[4] leg follows as the `config5` object); 5 = T=16, 256x256, 64 clips per GPU, '
                         'TSM_DTYPE_BF16 as the headline itself (sets --dtype/--batch/--segments/--size, no alt / config5 / CPU legs)')
    args = ap.parse_args()
    if args.config == 5:
        args.dtype, args.batch, args.segments, args.size = 'bf16', 64, 16, 256
        args.no_alt = args.no_cpu_baseline = args.no_config5 = True
    headline_defaults = (args.dtype, args.batch, args.segments, args.size) == ('f32', 32, 8, 224)
    if not headline_defaults:
        args.no_config5 = True       # the extra leg rides behind the driver's default command only

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # the driver's command form for N > 1 may come without a launcher: start the ranks ourselves, BEFORE torch
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback for the product path)'
    # Rehearsal hook for boxes with fewer GPUs than ranks (TSM_BENCH_REHEARSAL=1): all ranks share cuda:0 and
    # the process group is gloo.  Exercises the N > 1 control flow only; its numbers mean nothing.
    rehearsal = os.environ.get('TSM_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # TSM_BENCH_FORCE_COLLECTIVE=1 on ONE GPU: bring up a 1-rank nccl (RCCL) group and keep the all-gather inside the
    # step, so that the exact code path of an N > 1 run (device_id= init, device-tensor collective) executes on a
    # single-GPU box.  The number is still a 1-GPU number and says so in config.parallelism.
    forced = world == 1 and os.environ.get('TSM_BENCH_FORCE_COLLECTIVE') == '1'
    if world > 1 or forced:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if forced:
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    collective = world > 1 or forced
    backend = dist.get_backend() if collective else None
    # One tile-tuning pass per job, not per rank: every rank points at the same TSM_TUNE_CACHE file (self_launch sets
    # it; under an external launcher rank 0 names one here), rank 0 creates + tunes its engine first, the others
    # read its choices.  Tile choices never change a result bit, but ranks on different choices would time
    # different kernels.
    if world > 1:
        name = [os.environ.get('TSM_TUNE_CACHE')]
        if name[0] is None and rank == 0:
            import tempfile
            name[0] = os.path.join(tempfile.mkdtemp(prefix='tsm_bench_'), 'tune_cache.txt')
        dist.broadcast_object_list(name, src=0, device=torch.device('cpu') if rehearsal else torch.device('cuda', local_rank))
        os.environ['TSM_TUNE_CACHE'] = name[0]
        tune_note = 'rank 0 tuned in this job, the other ranks read its choices from a job-private TSM_TUNE_CACHE file'
    elif 'TSM_TUNE_CACHE' not in os.environ:
        # A single-GPU bench never reads the per-user tune cache: the first process to tune a build may have been a profiler
        # run (serialised dispatches favour the one-launch forms) and its choices would silently be the headline's.
        os.environ['TSM_TUNE_CACHE'] = 'off'
        tune_note = 'tuned in this process (TSM_TUNE_CACHE=off: the per-user tune cache is neither read nor written)'
    else:
        tune_note = 'TSM_TUNE_CACHE=%s from the environment' % os.environ['TSM_TUNE_CACHE']

    from workoutdetector_amd.build import build_library
    if rank == 0:
        build_library()
    if collective:
        dist.barrier()
    from workoutdetector_amd import _lib
    from workoutdetector_amd import distributed as tdist
    from workoutdetector_amd.distributed import all_gather_logits
    from workoutdetector_amd.engine import TsmEngine
    from workoutdetector_amd.flops import layer_table
    tdist.set_force_collective(forced)
    from workoutdetector_amd.weights import make_state_dict
    build_id = _lib.load().tsm_build_id().decode()

    sd = make_state_dict(0, 12)

    def make_clips(B, T, H, W):
        gen = torch.Generator(device='cuda').manual_seed(rank)
        return torch.randn(B, T, 3, H, W, device='cuda', generator=gen)

    def run_mode(dtype, want_launch_times, geom, clips, parity_idx):
        """W warm-up + K timed steps of one engine on `geom` = (T, H, W, B); returns a dict: wall seconds (max over ranks),
        event timings, the tuner's tiles and the logits of the last timed step at `parity_idx`."""
        T, H, W, B = geom
        if world > 1 and rank != 0:
            dist.barrier()       # rank 0 is tuning; its choices are in TSM_TUNE_CACHE when this returns
        eng = TsmEngine(num_class=12, num_segments=T, height=H, width=W, max_clips=B, device=local_rank,
                        state_dict=sd, dtype=dtype)
        logits = torch.empty(B, 12, device='cuda')
        eng.warmup([B])      # one-time tile / split-K autotuning of this batch size: initialisation, never a timed step
        if world > 1 and rank == 0:
            dist.barrier()

        def step():
            eng.forward_device(clips, out=logits)
            return all_gather_logits(logits) if collective else logits

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        n_timed = min(args.steps, 64)
        if want_launch_times:
            eng.set_layer_timing(n_timed, only_conv3x3=True)   # HIP-event pairs around the dominant kernel's launches
        # one event per step boundary on the stream the steps run on (torch's current stream is the stream handed to
        # tsm_forward): per-step durations without a host sync inside the timed region
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            out = step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        assert bool(torch.isfinite(out).all())
        res = {'dtype': dtype, 'geom': geom}
        # the timed steps' own output, kept for the parity check below (a [world * B, 12] gather starts with this rank's rows)
        res['timed_logits'] = out[:B][parity_idx].detach().float().cpu().numpy()
        # Per-launch durations of the timed forwards (events were recorded inside the timed region; reading
        # them here keeps the host syncs out of it).
        res['per_launch'] = [eng.layer_times_ms(i) for i in range(n_timed)] if want_launch_times else []
        res['step_ms'] = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        eng.set_layer_timing(0)
        # Whole-forward kernel time (one HIP-event pair around all launches of a forward), outside the
        # wall-clock region because reading it synchronises.
        fwd_ev_ms = []
        for _ in range(min(args.steps, 10)):
            eng.forward_device(clips, out=logits)
            fwd_ev_ms.append(eng.last_forward_ms)
        torch.cuda.synchronize()
        t_max = torch.tensor([elapsed], device='cuda')
        if collective:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
            sm = res['step_ms']
            mine = torch.tensor([[sm[len(sm) // 2], sorted(fwd_ev_ms)[len(fwd_ev_ms) // 2]]], device='cuda')
            res['rank_ms'] = all_gather_logits(mine).cpu().tolist()      # [world][step median, forward-kernel median]
            # the exchange step on its own ("all-gather us"), outside the timed region
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                all_gather_logits(logits)
            torch.cuda.synchronize()
            res['exchange_us'] = 1e6 * (time.perf_counter() - t0) / 20
        res['tiles'] = eng.conv_tiles(B)
        eng.close()
        res['elapsed'] = float(t_max.item())
        res['fwd_ms'] = sorted(fwd_ev_ms)[len(fwd_ev_ms) // 2]
        return res

    def roofline_of(res):
        """The `roofline` object of one timed mode: the dominant kernel (the instantiation that runs most of the 3x3 time of
        layer2-4) priced by its algorithmic flops and live launch durations, the whole forward beside it, PMC traffic from
        the committed stamped passes."""
        T, H, W, B = res['geom']
        dtype, tiles, per_launch, fwd_ms = res['dtype'], res['tiles'], res['per_launch'], res['fwd_ms']
        gflop = flops_per_clip(T, H, W) / 1e9
        fwd_achieved = gflop * B / fwd_ms  # GFLOP / ms == TFLOP/s
        frames = B * T
        table = {r['name']: r for r in layer_table(H, W)}
        dom = [r for r in table.values() if r['k'] == 3 and r['s'] >= 1 and not r['name'].startswith('layer1.')]
        # The engine tunes the tile shape (and the conv2 + conv3 fusion) per layer, so the 3x3 convs may run on more
        # than one kernel: the dominant kernel is the one with the most time.  A fused launch is priced with the
        # work it really does (its block's conv3 on top of the 3x3) and never mixed into the plain-3x3 group.
        groups = {}
        for r in dom:
            if '+conv2' in tiles.get(r['name'].replace('.conv2', '.conv1'), ''):
                continue      # layer2.0's 3x3 rode in its conv1's launch (front_s2_kernel): no launch of its own, no time of its own
            kern, with_conv3 = kernel_of(tiles[r['name']], dtype, r['cout'], r['s'])
            gf = 2.0 * r['macs'] * frames / 1e9
            if with_conv3:
                gf += 2.0 * table[r['name'].replace('.conv2', '.conv3')]['macs'] * frames / 1e9
            g = groups.setdefault(kern, {'ms': [], 'gflop': []})
            for d in per_launch:
                g['ms'].append(d[r['name']])
                g['gflop'].append(gf)
        dom_kernel, g = max(groups.items(), key=lambda kv: sum(kv[1]['ms']))
        dom_ms = g['ms']
        dom_gflop = sum(g['gflop']) / len(g['gflop'])
        peak = PEAK_F32_MFMA_TFLOPS if dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS
        peak_name = ('dense bf16 MFMA (v_mfma_f32_32x32x16_bf16); the kernel executes 3 MFMA FLOPs per algorithmic FLOP'
                     if dtype == 'bf16x3' else 'dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)' if dtype == 'bf16'
                     else 'exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), dense')
        dom_avg_ms = sum(dom_ms) / len(dom_ms)
        achieved = dom_gflop / dom_avg_ms
        traffic_entry, traffic_stale = measured_traffic(B, T, H, W, dom_kernel)
        roof = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak,
                'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                'traffic': (traffic_entry or {}).get('hbm_bytes_per_launch'),
                **({'traffic_stale': traffic_stale} if traffic_stale else {}),
                'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/traffic.json)',
                'kernel': dom_kernel + ' (3x3 convs of layer2-4%s, %d of 13 launches per forward)'
                          % (', each with its block\'s conv3 + residual fused behind it'
                             if dom_gflop > 1.01 * 2.0 * dom[0]['macs'] * frames / 1e9 else '',
                             len(dom_ms) // len(per_launch)),
                'gflop_per_launch': round(dom_gflop, 3), 'avg_launch_ms': round(dom_avg_ms, 4),
                'launches_timed': len(dom_ms),
                'peak_name': peak_name,
                'forward_achieved': round(fwd_achieved, 2),
                'forward_frac': round(fwd_achieved / peak, 4),
                'forward_gflop': round(gflop * B, 3), 'forward_kernel_ms': round(fwd_ms, 4)}
        if traffic_entry and traffic_entry.get('forward_hbm_bytes'):
            # whole-forward HBM rate from the committed PMC passes (bytes) over the live forward time: the second
            # roofline SURVEY 8d asks for beside the MFMA one (matters for the bf16 formats)
            gbs = traffic_entry['forward_hbm_bytes'] / 1e9 / (fwd_ms / 1e3)
            roof.update({'forward_hbm_bytes': traffic_entry['forward_hbm_bytes'], 'forward_hbm_gbs': round(gbs, 1),
                         'forward_hbm_frac': round(gbs / PEAK_HBM_GBS, 4), 'hbm_peak_gbs': PEAK_HBM_GBS})
        return roof

    def workload_of(geom, dtype, config_index):
        T, H, W, B = geom
        return (f'TSM-R50 {T}-seg {H}x{W} 12-class inference, batch {B} clips per GPU, ' +
                '%s NHWC, device-resident input (BASELINE.json configs[%d])' % (dtype, config_index))

    geom = (args.segments, args.size, args.size, args.batch)
    T, H, W, B = geom
    clips = make_clips(B, T, H, W)
    parity_idx = sorted({0, B // 2, B - 1})[:PARITY_CLIPS]
    main_res = run_mode(args.dtype, True, geom, clips, parity_idx)
    runs = {args.dtype: main_res}
    alt_res = None
    if not args.no_alt:
        alt_dtype = 'bf16x3' if args.dtype == 'f32' else 'f32'
        alt_res = runs[alt_dtype] = run_mode(alt_dtype, False, geom, clips, parity_idx)

    parity = {}
    if rank == 0 and not args.no_parity:
        # Parity of what was TIMED, in this process: the CPU oracle on a few of the timed clips (outside the timed region)
        want = oracle_logits(sd, clips[parity_idx].cpu(), T, list(runs))
        parity = {d: parity_of(r['timed_logits'], want, d) for d, r in runs.items()}
        for d in parity:
            parity[d]['clips'] = parity_idx
    del clips

    # BASELINE configs[4] behind the headline: the stress shape (bf16 weights + activations, T = 16, 256 x 256, 64 clips per
    # GPU) through the same call (utils/inference_count.py:273-275 / models/tsm.py:409-419), its own tuning, the same
    # steps / warm-up, its own roofline and its own parity check (the bf16-storage oracle, on the first and the last clip).
    c5_res, c5_parity = None, None
    if not args.no_config5:
        geom5 = (16, 256, 256, 64)
        clips5 = make_clips(64, 16, 256, 256)
        idx5 = [0, 63]
        c5_res = run_mode('bf16', True, geom5, clips5, idx5)
        if rank == 0 and not args.no_parity:
            want5 = oracle_logits(sd, clips5[idx5].cpu(), 16, ['bf16'])
            c5_parity = parity_of(c5_res['timed_logits'], want5, 'bf16')
            c5_parity['clips'] = idx5
        del clips5

    if rank == 0:
        elapsed = main_res['elapsed']
        clips_total = B * world * args.steps
        value = clips_total / elapsed
        gflop = flops_per_clip(T, H, W) / 1e9
        sm = main_res['step_ms']
        if rehearsal:
            parallelism = (f'REHEARSAL: {world} ranks sharing cuda:0, gloo all-gather through host memory -- control flow '
                           'only, not a measurement')
        elif world > 1:
            parallelism = f'clip-sharded x{world}, RCCL (nccl backend) all-gather of logits'
        elif forced:
            parallelism = 'single GPU; 1-rank RCCL (nccl backend) group, all-gather of logits kept inside the step'
        else:
            parallelism = 'single GPU'
        line = {
            'metric': f'clips/sec ({T}x3x{H}x{W} TSM-R50)', 'value': None if rehearsal else round(value, 2),
            'unit': 'clips/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'step_ms': {'min': round(sm[0], 4), 'median': round(sm[len(sm) // 2], 4), 'max': round(sm[-1], 4),
                        'note': 'rank-0 per-step durations between HIP events on the step stream: the noise floor '
                                'design decisions must be read against'},
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload_of(geom, args.dtype, 4 if args.config == 5 else 1),
                       'clips_per_gpu': B, 'num_segments': T, 'height': H, 'width': W, 'num_class': 12,
                       'weights': 'seeded random init (no trained weights offline)',
                       'parallelism': parallelism},
            'roofline': roofline_of(main_res),
            'build_id': build_id,
            'tune': tune_note,
        }
        if parity:
            line['parity'] = parity[args.dtype]
        if rehearsal:
            line['rehearsal'] = True
        if collective:
            line['exchange'] = {'collective': 'all_gather_into_tensor of f32[%d, 12] per rank (backend %s%s)'
                                              % (B, backend, ' = RCCL' if backend == 'nccl' else ': NOT RCCL, rehearsal'),
                                'avg_us': round(main_res['exchange_us'], 1),
                                'per_rank_step_ms_median': [round(r[0], 4) for r in main_res['rank_ms']],
                                'per_rank_forward_kernel_ms_median': [round(r[1], 4) for r in main_res['rank_ms']],
                                'tune_cache': 'one TSM_TUNE_CACHE file for all ranks: rank 0 tunes, the others read its '
                                              'tile choices' if world > 1 else 'single rank',
                                'note': 'back-to-back latency of the only data-path collective, measured outside the timed steps'}
        if alt_res is not None:
            a_dtype, a_elapsed, a_fwd = alt_res['dtype'], alt_res['elapsed'], alt_res['fwd_ms']
            line['alt_precision'] = {
                'dtype': a_dtype, 'value': None if rehearsal else round(clips_total / a_elapsed, 2), 'unit': 'clips/s',
                'ms_per_step': round(1e3 * a_elapsed / args.steps, 4), 'forward_kernel_ms': round(a_fwd, 4),
                'forward_algorithmic_tflops': round(gflop * B / a_fwd, 2),
                'note': ('same engine, same inputs, same steps/warmup, TSM_DTYPE_BF16X3: split-bf16 storage (hi/lo), '
                         'a*b = ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; passes the '
                         'same parity tests (logits within 5e-6 of the fp32 oracle, bar 1e-3; tests/test_bf16x3_gpu.py); '
                         'not the headline because it is not bit-level fp32 arithmetic')
                        if a_dtype == 'bf16x3' else 'exact-fp32 MFMA mode of the same engine'}
            if a_dtype in parity:
                line['alt_precision']['parity'] = parity[a_dtype]
        if c5_res is not None:
            c5_total = 64 * world * args.steps
            c5_sm = c5_res['step_ms']
            line['config5'] = {
                'metric': 'clips/sec (16x3x256x256 TSM-R50)',
                'value': None if rehearsal else round(c5_total / c5_res['elapsed'], 2), 'unit': 'clips/s',
                'dtype': 'bf16', 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(1e3 * c5_res['elapsed'] / args.steps, 4),
                'step_ms': {'min': round(c5_sm[0], 4), 'median': round(c5_sm[len(c5_sm) // 2], 4), 'max': round(c5_sm[-1], 4)},
                'config': {'workload': workload_of((16, 256, 256, 64), 'bf16', 4), 'clips_per_gpu': 64, 'num_segments': 16,
                           'height': 256, 'width': 256, 'num_class': 12},
                'roofline': roofline_of(c5_res),
                'note': 'BASELINE.json configs[4] (the MFMA / HBM stress shape) on this rank count: its own engine, tuning, '
                        'timed steps and parity check, run behind the headline in the same process; NOT the metric the '
                        'headline `value` is quoted on'}
            if collective:
                line['config5']['exchange_avg_us'] = round(c5_res['exchange_us'], 1)
            if c5_parity is not None:
                line['config5']['parity'] = c5_parity
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sd, T, H, W)
        print(json.dumps(line), flush=True)
        bad = [d for d, q in parity.items() if not q['ok']]
        if c5_parity is not None and not c5_parity['ok']:
            bad.append('config5/bf16')
        if bad:
            raise SystemExit(f'parity check of the timed logits FAILED for {bad}: '
                             f'{[parity[d] if d in parity else c5_parity for d in bad]}')
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
