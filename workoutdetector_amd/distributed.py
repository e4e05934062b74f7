"""Multi-GPU sharding of the clip stream: one process per GPU, clips are the independent unit.

The reference's inference path is a single-process batch-1 loop (utils/inference_count.py:411-416)
with no collective.  Clips never interact (the temporal shift stays inside a clip), so clip ``j`` of
a job goes to rank ``j // ceil(n / W)`` (contiguous blocks keep order), every rank runs its own
engine with replicated weights, and the only exchange is ONE all-gather of per-clip logits
``float32[ceil(n/W), num_class]`` before the serial, host-side rep counter.
The payload is a few KB, so the step is latency-bound; xGMI bandwidth is irrelevant here.

Backend: ``nccl`` (= RCCL over xGMI on ROCm) for CUDA tensors, ``gloo`` for the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


_force_collective = False


def world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def set_force_collective(on: bool) -> None:
    """With a process group of ONE rank the exchange step is the identity and is normally skipped.  ``True`` runs the
    collectives anyway (pad -> all-gather -> trim on the group's backend): this is how a single-GPU box exercises the
    real ``nccl`` (= RCCL) branch -- device-tensor all-gathers, ``device_id=`` initialisation -- before a multi-GPU
    node does (tests/test_rccl_gpu.py, ``bench.py`` under TSM_BENCH_FORCE_COLLECTIVE=1)."""
    global _force_collective
    _force_collective = bool(on)


def collective_enabled() -> bool:
    """True when the data-path collectives must run: more than one rank, or forced on an initialised group."""
    rank, world = world_info()
    return world > 1 or (_force_collective and dist.is_available() and dist.is_initialized())


def on_rccl() -> bool:
    """The process group moves CUDA tensors (backend ``nccl`` = RCCL over xGMI on ROCm)."""
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'


def per_rank(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """[lo, hi) of the contiguous block of items owned by ``rank`` (may be empty at the tail)."""
    per = per_rank(n_items, world)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def all_gather_logits(local: torch.Tensor, group=None) -> torch.Tensor:
    """[per, C] on every rank (same ``per`` everywhere) -> [W*per, C] in rank order, on every rank."""
    rank, world = world_info()
    if not collective_enabled():
        return local
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def gather_clip_logits(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Ragged form: rank r holds the logits of its ``shard_range`` block (possibly fewer than
    ``per_rank`` rows, possibly none).  Pads to ``per_rank`` rows, all-gathers once, trims to
    ``n_total`` rows in global clip order."""
    rank, world = world_info()
    if not collective_enabled():
        return local[:n_total]
    per = per_rank(n_total, world)
    lo, hi = shard_range(n_total, world, rank)
    assert local.shape[0] == hi - lo, f'rank {rank}: got {local.shape[0]} rows for block [{lo},{hi})'
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    return all_gather_logits(local, group)[:n_total]


def shard_list(items: List, world: int, rank: int) -> List:
    lo, hi = shard_range(len(items), world, rank)
    return items[lo:hi]


# ---- dataset-level sharding: whole videos over ranks, balanced by clip count -------------------------------------
def plan_video_shards(clip_counts: Sequence[int], world: int) -> List[int]:
    """Owner rank of every video for a dataset run (BASELINE config 4): longest-processing-time-first -- videos in
    order of decreasing (estimated) clip count, each to the rank with the fewest clips so far (ties: lowest video
    index / lowest rank).  Deterministic, so every rank computes the same plan from the same counts with no exchange.
    The counts only steer the balance; a wrong estimate costs speed, never correctness."""
    load = [0] * world
    owner = [0] * len(clip_counts)
    for v in sorted(range(len(clip_counts)), key=lambda i: (-int(clip_counts[i]), i)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[v] = r
        load[r] += int(clip_counts[v])
    return owner


def shard_efficiency(clip_counts: Sequence[int], owner: Sequence[int], world: int) -> float:
    """Modelled strong-scaling efficiency of a plan whose ranks run independently until ONE final exchange:
    (total clips / W) / (clips of the busiest rank)."""
    load = [0] * world
    for c, r in zip(clip_counts, owner):
        load[r] += int(c)
    return (sum(load) / world) / max(load) if max(load) > 0 else 1.0


def lockstep_efficiency(clip_counts: Sequence[int], world: int) -> float:
    """The same figure for round-robin rounds of W videos with an exchange per round (``shard='videos'``): every round
    lasts as long as its longest video."""
    total = sum(int(c) for c in clip_counts)
    span = sum(max(int(c) for c in clip_counts[r0:r0 + world]) for r0 in range(0, len(clip_counts), world))
    return (total / world) / span if span > 0 else 1.0
