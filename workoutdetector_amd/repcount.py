"""RepCount annotation helper: enumeration of videos and the count metric, either side of the hot path.

Counterpart of ``RepcountHelper.get_rep_data`` / ``eval_count`` and ``RepcountItem``
(workoutdetector/datasets/repcount_dataset.py:115-138,169-251).  Only the parts
``inference_dataset`` and the evaluation need; the training dataset classes are out of scope.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import pandas as pd

CLASSES = ['situp', 'push_up', 'pull_up', 'jump_jack', 'squat', 'front_raise']


@dataclass
class RepcountItem:
    video_path: str
    frames_path: str
    total_frames: int
    class_: str
    count: int
    reps: List[int]           # start_1, end_1, start_2, end_2, ...
    split: str
    video_name: str
    ytb_id: Optional[str] = None
    ytb_start_sec: Optional[float] = None
    ytb_end_sec: Optional[float] = None

    def __getitem__(self, key):
        return self.__dict__[key]


class RepcountHelper:
    """``data_root`` holds ``videos/{split}/{name}`` (and optionally ``rawframes/{split}/{stem}/``);
    ``anno_file`` is RepCount's ``annotation.csv`` (cols: class_, split, name, vid, start, end, count, reps)."""

    def __init__(self, data_root: str, anno_file: str):
        self.data_root = data_root
        self.anno_file = anno_file
        self.classes = list(CLASSES)  # bench_pressing is excluded by the reference (uncleaned labels)

    def get_rep_data(self, split: Sequence[str] = ('test',), action: Sequence[str] = ('situp',)) -> Dict[str, RepcountItem]:
        assert len(split) > 0, 'split must be specified, e.g. ["train", "val"]'
        assert len(action) > 0, 'action must be specified, e.g. ["pull_up", "squat"]'
        split = [s.lower() for s in split]
        action = [a.lower() for a in action]
        if 'all' in action:
            action = self.classes
        df = pd.read_csv(self.anno_file, index_col=0)
        df = df[df['split'].isin(split) & df['class_'].isin(action)].reset_index(drop=True)
        out: Dict[str, RepcountItem] = {}
        for _, row in df.iterrows():
            name = row['name']
            stem = name.split('.')[0]
            frames_path = os.path.join(self.data_root, 'rawframes', row['split'], stem)
            total = len(os.listdir(frames_path)) if os.path.isdir(frames_path) else -1
            count = int(row['count'])
            reps = [int(x) for x in str(row['reps']).split()] if count > 0 else []
            out[name] = RepcountItem(os.path.join(self.data_root, 'videos', row['split'], name), frames_path, total,
                                     row['class_'], count, reps, row['split'], name, row['vid'], row['start'],
                                     row['end'])
        return out

    def eval_count(self, pred_counts: Dict[str, int], split: Sequence[str] = ('test',),
                   action: Sequence[str] = ('all',)) -> Tuple[float, float, Dict[str, dict]]:
        """(mean(|diff| / gt) with 0 for gt == 0, fraction with |diff| <= 1, per-video records),
        both averaged over ALL items of the split/action selection, like the reference."""
        items = self.get_rep_data(split=split, action=action)
        tot_mae, tot_obo = 0.0, 0.0
        per: Dict[str, dict] = {}
        for name, c in pred_counts.items():
            gt = items[name].count
            diff = abs(c - gt)
            mae = diff / gt if gt > 0 else 0
            obo = diff <= 1
            tot_mae += mae
            tot_obo += obo
            per[name] = dict(pred_count=c, gt_count=gt, mae=mae, obo_acc=obo)
        return tot_mae / len(items), tot_obo / len(items), per
