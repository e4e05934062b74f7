"""Evaluate score files: scores -> states -> counts -> MAE / OBO.

Counterpart of workoutdetector/utils/eval.py: ``main`` :117-180 (softmax option, arg-max,
``score >= 0.5`` else -1, ``pred_to_count(step=8)``, ``obo_mae``), ``obo_mae`` :11-24 and
``analyze_count(csv, out_csv)`` :58-114 (same signature, CSV schema and arithmetic).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import pandas as pd

from .counting import obo_mae, pred_to_count, to_softmax  # noqa: F401  (re-export)

THRESHOLD = 0.5
STEP = 8


def preds_from_scores(scores: Dict[str, Dict[str, float]], softmax: bool = False,
                      threshold: float = THRESHOLD) -> List[int]:
    """One state per clip in file order: first arg-max class (``max`` over dict items keeps the first
    of ties), kept if its score >= threshold, else -1 (utils/eval.py:153-164)."""
    preds = []
    for v in scores.values():
        if softmax:
            v = to_softmax(v)
        class_id, score = max(v.items(), key=lambda kv: kv[1])
        preds.append(int(class_id) if score >= threshold else -1)
    return preds


def evaluate_dir(json_dir: str, anno_path: str, softmax: bool = False) -> Tuple[pd.DataFrame, float, float]:
    files = sorted(f for f in os.listdir(json_dir) if f.endswith('.json'))
    anno = pd.read_csv(anno_path, index_col='name')
    rows, preds, gts = [], [], []
    for f in files:
        video_name = f.split('.')[0] + '.mp4'
        with open(os.path.join(json_dir, f)) as fp:
            data = json.load(fp)
        pred = preds_from_scores(data['scores'], softmax=softmax)
        count, reps = pred_to_count(pred, step=STEP)
        gt_count = int(anno.loc[video_name]['count'])
        preds.append(count)
        gts.append(gt_count)
        rows.append([video_name, gt_count, count, anno.loc[video_name]['reps'], reps,
                     anno.loc[video_name]['split'], data['action']])
    mae, obo = obo_mae(preds, gts) if preds else (float('nan'), float('nan'))
    df = pd.DataFrame(rows, columns=['name', 'gt_count', 'pred_count', 'gt_rep', 'pred_rep', 'split', 'action'])
    return df, mae, obo


def main(json_dir: str, anno_path: str, out_csv: Optional[str], softmax: bool = False) -> Tuple[float, float]:
    df, mae, obo = evaluate_dir(json_dir, anno_path, softmax)
    if out_csv:
        df.to_csv(out_csv)
        print(f'Done. csv file saved to {out_csv}')
    print(f'=====Mean absolute error: {mae:.4f}, OBO acc: {obo:.4f}=====')
    return mae, obo


def analyze_count(csv: str, out_csv: Optional[str]) -> None:
    """Per-(split, action) summary of an evaluation CSV: drop-in for utils/eval.py:58-114.

    ``csv`` is what ``main`` writes (columns ``,name,gt_count,pred_count,gt_rep,pred_rep,split,action``); the
    result has columns ``,action,split,mae,obo_acc,total,avg_count`` -- one row per (split, action) in order of first
    appearance, splits outermost, then one ``action == 'all'`` row per split -- and goes to ``out_csv`` (if given)
    and to stdout, like the reference.  Reference arithmetic kept as is: ``obo_acc`` is the COUNT of videos with
    ``|pred - gt| == 1`` (``obo_mae(ratio=False)``), not a fraction; the per-split MAE re-accumulates
    ``int(mae * n)`` per action (truncating); a (split, action) pair without videos raises ZeroDivisionError."""
    df = pd.read_csv(csv, index_col='name')
    actions, splits = df.action.unique(), df.split.unique()
    rows = []
    per_split = {sp: dict(mae=0, obo=0, total=0, avg_count=0.0) for sp in splits}
    for sp in splits:
        acc = per_split[sp]
        for act in actions:
            sel = df.loc[(df.action == act) & (df.split == sp)]
            gt, pred = sel.gt_count.values, sel.pred_count.values
            mae, obo = obo_mae(pred, gt, ratio=False)
            rows.append([act, sp, mae, obo, len(sel), np.mean(gt)])
            acc['mae'] += int(mae * len(sel))
            acc['obo'] += int(obo)
            acc['total'] += len(sel)
            acc['avg_count'] += gt.sum()
    out = pd.DataFrame(rows, columns=['action', 'split', 'mae', 'obo_acc', 'total', 'avg_count'])
    for sp in splits:
        acc = per_split[sp]
        print(f'{sp}: {acc}')
        n = acc['total']
        out = pd.concat([out, pd.DataFrame({'action': 'all', 'split': sp, 'mae': acc['mae'] / n, 'obo_acc': acc['obo'],
                                            'total': n, 'avg_count': acc['avg_count'] / n}, index=[0])],
                        ignore_index=True)
    if out_csv:
        out.to_csv(out_csv)
    print(out)


def summarize_counts(df: pd.DataFrame) -> pd.DataFrame:
    """Not in the reference: per (split, action) videos, MAE (un-normalised), OBO as a FRACTION (|diff| == 1) and
    mean |diff| / gt, from an in-memory frame (the helper this module called ``analyze_count`` before round 2)."""
    out = []
    for (split, action), g in df.groupby(['split', 'action']):
        diff = (g['pred_count'] - g['gt_count']).abs()
        out.append(dict(split=split, action=action, n=len(g), mae=float(diff.mean()), obo=float((diff == 1).mean()),
                        mae_norm=float((diff / g['gt_count'].clip(lower=1)).mean())))
    return pd.DataFrame(out)
