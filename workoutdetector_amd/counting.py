"""Scores -> states -> repetition count: the host-side post-step of the hot path.

Counterparts of (reference file:line):
  pred_to_count     workoutdetector/utils/inference_count.py:114-165
  to_softmax        workoutdetector/utils/visualize.py:140-150
  scores_to_preds   workoutdetector/utils/eval.py:153-164 (arg-max, ``score >= threshold`` else -1)
  obo_mae           workoutdetector/utils/eval.py:11-24
  eval_count        workoutdetector/datasets/repcount_dataset.py:212-251 (metric part)

The counter is an O(n_clips) integer state machine with a serial dependency, so it stays on the host
(SURVEY.md section 8a row A8); it consumes the logits the HIP engine produced.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np


class RepCounter:
    """Incremental form of ``pred_to_count``: feed one state per clip, read ``count``/``reps`` any time.
    ``push`` over a whole list gives exactly ``pred_to_count(list, step)``."""

    def __init__(self, step: int = 8):
        self.step = step
        self.count = 0
        self.reps: List[int] = []
        self._n = 0            # clips seen so far (including -1)
        self._last: Optional[int] = None    # last kept (non -1) state
        self._start = 0        # index where the current run started
        self._start_val: Optional[int] = None   # raw prediction at _start (may be -1)

    def push(self, pred: int) -> int:
        idx = self._n
        self._n += 1
        if idx == 0:
            self._start_val = pred
        if pred == -1:
            return self.count
        if self._last is not None and self._last != pred and pred % 2 == 1 and self._last == pred - 1:
            self.count += 1
            self.reps.extend((self._start * self.step, idx * self.step))
        self._last = pred
        if pred != self._start_val:
            self._start, self._start_val = idx, pred
        return self.count


def pred_to_count(preds: Sequence[int], step: int) -> Tuple[int, List[int]]:
    """Repetition count and [start_1, end_1, start_2, end_2, ...] (frame units) from per-clip states.

    A repetition is a kept transition 2k -> 2k+1 (start state -> end state of action k); ``-1``
    (below threshold) is skipped but still occupies an index."""
    rc = RepCounter(step)
    for p in preds:
        rc.push(int(p))
    assert rc.count * 2 == len(rc.reps)
    return rc.count, rc.reps


def softmax_rows(scores: np.ndarray) -> np.ndarray:
    """float32 softmax over the class axis (``F.softmax(torch.Tensor(values), dim=0)`` per clip)."""
    s = np.asarray(scores, dtype=np.float32)
    e = np.exp(s - s.max(axis=-1, keepdims=True), dtype=np.float32)
    return e / e.sum(axis=-1, keepdims=True, dtype=np.float32)


def to_softmax(d: Mapping[str, float]) -> Dict[str, float]:
    vals = softmax_rows(np.array(list(d.values()), dtype=np.float32))
    return dict(zip(d.keys(), vals))


def scores_to_preds(scores: Iterable[Sequence[float]], threshold: float = 0.5, softmax: bool = True) -> List[int]:
    """Per clip: (softmax,) first arg-max, class id if its score >= threshold else -1."""
    arr = np.asarray(list(scores), dtype=np.float32)
    if arr.size == 0:
        return []
    p = softmax_rows(arr) if softmax else arr
    best = p.argmax(axis=1)           # first maximum, like Python's max() over dict items
    top = p[np.arange(len(p)), best]
    return [int(b) if t >= threshold else -1 for b, t in zip(best, top)]


def obo_mae(preds: Sequence[int], targets: Sequence[int], ratio: bool = True):
    """eval.py metric: un-normalised MAE; OBO counts ``abs(diff) == 1`` only."""
    mae = 0.0
    obo = 0.0
    for p, t in zip(preds, targets):
        mae += abs(p - t)
        obo += (abs(p - t) == 1)
    n = len(preds)
    return (mae / n, obo / n) if ratio else (mae / n, obo)


def eval_count(pred_counts: Mapping[str, int], gt_counts: Mapping[str, int]) -> Tuple[float, float]:
    """RepcountHelper.eval_count metric: MAE = mean(|diff| / gt) (0 for gt == 0), OBO = |diff| <= 1,
    both averaged over the ground-truth items."""
    tot_mae, tot_obo = 0.0, 0.0
    for name, c in pred_counts.items():
        gt = gt_counts[name]
        diff = abs(c - gt)
        tot_mae += diff / gt if gt > 0 else 0
        tot_obo += (diff <= 1)
    n = len(gt_counts)
    return tot_mae / n, tot_obo / n
