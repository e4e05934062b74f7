"""Evaluate score files: scores -> states -> counts -> MAE / OBO.

Counterpart of workoutdetector/utils/eval.py: ``main`` :117-180 (softmax option, arg-max,
``score >= 0.5`` else -1, ``pred_to_count(step=8)``, ``obo_mae``), ``obo_mae`` :11-24 and the
per-action summary of ``analyze_count`` :58-114.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

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


def analyze_count(df: pd.DataFrame) -> pd.DataFrame:
    """Per (split, action): videos, MAE (un-normalised), OBO (|diff| == 1), mean |diff|/gt."""
    out = []
    for (split, action), g in df.groupby(['split', 'action']):
        diff = (g['pred_count'] - g['gt_count']).abs()
        out.append(dict(split=split, action=action, n=len(g), mae=float(diff.mean()), obo=float((diff == 1).mean()),
                        mae_norm=float((diff / g['gt_count'].clip(lower=1)).mean())))
    return pd.DataFrame(out)
