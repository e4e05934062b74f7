"""CPU restatement of the post-step of the hot path: scores -> states -> repetition count.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, in /root/reference:
  pred_to_count       workoutdetector/utils/inference_count.py:146-165
  to_softmax          workoutdetector/utils/visualize.py:140-150  (softmax over the class values)
  scores_to_preds     workoutdetector/utils/eval.py:153-164       (arg-max, score >= 0.5 else -1)
  obo_mae             workoutdetector/utils/eval.py:11-24
  eval_count          workoutdetector/datasets/repcount_dataset.py:212-251
  clip_starts/window  workoutdetector/utils/inference_count.py:411-414

Pinned by tests/golden/counter_kats.json (the reference's own known-answer
vectors) and tests/golden/ref_pred_to_count.json (outputs of the reference's own
function body run in the build container).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple


def pred_to_count(preds: Sequence[int], step: int) -> Tuple[int, List[int]]:
    """Sequential state machine: count a repetition on each kept transition 2k -> 2k+1.

    ``-1`` entries are skipped but keep their index.  ``start`` is the index where the
    current run began; it is re-based whenever the kept prediction differs from the raw
    ``preds[start]`` (which may itself be -1 at index 0).
    """
    count = 0
    reps: List[int] = []
    last = None
    start = 0
    for idx, p in enumerate(preds):
        if p == -1:
            continue
        if last is not None and last != p and p % 2 == 1 and last == p - 1:
            count += 1
            reps.extend((start * step, idx * step))
        last = p
        if p != preds[start]:
            start = idx
    return count, reps


def softmax(values: Sequence[float]) -> List[float]:
    """float32-free, numerically stable softmax over one clip's class scores."""
    m = max(values)
    e = [math.exp(v - m) for v in values]
    s = sum(e)
    return [x / s for x in e]


def scores_to_preds(scores: Sequence[Sequence[float]], threshold: float = 0.5,
                    use_softmax: bool = True) -> List[int]:
    """Per clip: optional softmax, first arg-max (Python ``max`` keeps the first of ties),
    class id if its score >= threshold else -1."""
    out: List[int] = []
    for row in scores:
        r = softmax(row) if use_softmax else list(row)
        best = max(range(len(r)), key=lambda i: r[i])
        out.append(best if r[best] >= threshold else -1)
    return out


def obo_mae(preds: Sequence[int], targets: Sequence[int]) -> Tuple[float, float]:
    """eval.py definition: un-normalised MAE, OBO counts |diff| == 1 only."""
    n = len(preds)
    mae = sum(abs(p - t) for p, t in zip(preds, targets)) / n
    obo = sum(1 for p, t in zip(preds, targets) if abs(p - t) == 1) / n
    return mae, obo


def eval_count(pred_counts: Dict[str, int], gt_counts: Dict[str, int]) -> Tuple[float, float]:
    """RepcountHelper.eval_count definition: MAE = mean(|diff|/gt) (0 when gt == 0),
    OBO = fraction with |diff| <= 1; both divided by the number of ground-truth items."""
    tot_mae = 0.0
    tot_obo = 0
    for name, c in pred_counts.items():
        gt = gt_counts[name]
        diff = abs(c - gt)
        tot_mae += diff / gt if gt > 0 else 0
        tot_obo += diff <= 1
    n = len(gt_counts)
    return tot_mae / n, tot_obo / n


def clip_frame_indices(total_frames: int, start: int, span: int = 16, stride: int = 2) -> List[int]:
    """Indices of ``vid[start:start+span:stride]`` (<= 8 of them; the tail is zero-padded by the caller)."""
    return list(range(start, min(start + span, total_frames), stride))


def clip_starts(total_frames: int, step: int = 8) -> List[int]:
    return list(range(0, total_frames, step))
