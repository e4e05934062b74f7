"""Generate golden vectors by EXECUTING the reference's own function bodies (build container only).

The reference package cannot be imported here (module-level imports of cv2 / onnxruntime /
torchvision / mmaction / fvcore, all absent), but four dependency-light pieces of the hot path
only need ``torch`` + ``typing``.  This script reads the reference source files as text at run
time, pulls those definitions out with ``ast`` and executes them in a scratch namespace, then
stores INPUTS and OUTPUTS (data only -- no reference source text) as fixtures:

  ref_pred_to_count.json   pred_to_count          utils/inference_count.py:114-165
  ref_temporal_shift.npz   TemporalShift.shift    models/tsm.py:35-50
  ref_consensus.npz        SegmentConsensus       models/tsm.py:157-174
  ref_metrics.json         obo_mae / to_softmax   utils/eval.py:11-24, utils/visualize.py:140-150
  ref_analyze_count.json   analyze_count          utils/eval.py:58-114 (input CSV text -> output CSV text + stdout)
  ref_ckpt_remap.json      the checkpoint key remap inside create_model   models/tsm.py:451-473

/root/reference does not exist on the GPU box; tests only read the committed fixtures.
Run:  python tests/golden/make_reference_vectors.py
"""
import ast
import json
import os
import typing
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference/workoutdetector'
HERE = os.path.dirname(os.path.abspath(__file__))


def _extract(path, name, kind=ast.FunctionDef, inside=None):
    tree = ast.parse(open(path).read())
    body = tree.body
    if inside is not None:
        (cls,) = [n for n in body if isinstance(n, ast.ClassDef) and n.name == inside]
        body = cls.body
    (node,) = [n for n in body if isinstance(n, kind) and n.name == name]
    if inside is not None:
        node.decorator_list = []
    mod = ast.Module(body=[node], type_ignores=[])
    ns = dict(torch=torch, F=F, nn=torch.nn, OrderedDict=OrderedDict)
    ns.update({k: getattr(typing, k) for k in ('List', 'Tuple', 'Dict', 'Union', 'Optional')})
    ns['OrderedDict'] = typing.OrderedDict
    exec(compile(mod, path, 'exec'), ns)
    return ns[name]


def main():
    rng = np.random.default_rng(20221004)

    # ---- pred_to_count ------------------------------------------------------------------
    ref_p2c = _extract(f'{REF}/utils/inference_count.py', 'pred_to_count')
    cases = []
    for n in [0, 1, 2, 3, 5, 8, 13, 21, 40, 64, 135, 330]:
        for variant in range(6):
            if variant < 2:      # uniform over -1..11
                p = rng.integers(-1, 12, n)
            elif variant < 4:    # one action (classes 2k, 2k+1) with background gaps, runs of 1-4
                k = int(rng.integers(0, 6))
                p, cur = [], 2 * k
                while len(p) < n:
                    run = int(rng.integers(1, 5))
                    sym = -1 if rng.random() < 0.2 else cur
                    p += [sym] * run
                    if sym != -1:
                        cur = 2 * k + (1 - (cur - 2 * k))
                p = np.array(p[:n], dtype=np.int64)
            else:                # mostly one action, occasional wrong-class spikes
                k = int(rng.integers(0, 6))
                p = 2 * k + rng.integers(0, 2, n)
                spikes = rng.random(n) < 0.1
                p = np.where(spikes, rng.integers(-1, 12, n), p)
            for step in (8, 1):
                preds = [int(v) for v in p]
                count, reps = ref_p2c(preds, step)
                cases.append(dict(preds=preds, step=step, count=int(count), reps=[int(r) for r in reps]))
    json.dump(dict(source='workoutdetector/utils/inference_count.py:114-165 executed via ast',
                   cases=cases), open(f'{HERE}/ref_pred_to_count.json', 'w'))
    print('pred_to_count cases:', len(cases))

    # ---- TemporalShift.shift ------------------------------------------------------------
    ref_shift = _extract(f'{REF}/models/tsm.py', 'shift', inside='TemporalShift')
    out = {}
    meta = []
    for i, (nb, t, c, h, w, div) in enumerate([(1, 8, 64, 3, 5, 8), (2, 8, 256, 2, 2, 8), (3, 4, 16, 1, 7, 8),
                                                (1, 16, 32, 2, 3, 8), (2, 8, 24, 2, 2, 3), (1, 1, 16, 2, 2, 8),
                                                (2, 2, 8, 3, 3, 8), (1, 8, 2048, 1, 1, 8)]):
        x = torch.from_numpy(rng.standard_normal((nb * t, c, h, w)).astype(np.float32))
        y = ref_shift(x.clone(), t, fold_div=div, inplace=False)
        out[f'x{i}'] = x.numpy()
        out[f'y{i}'] = y.numpy()
        meta.append([nb, t, c, h, w, div])
    out['meta'] = np.array(meta, dtype=np.int64)
    np.savez_compressed(f'{HERE}/ref_temporal_shift.npz', **out)
    print('temporal_shift cases:', len(meta))

    # ---- SegmentConsensus ('avg') -------------------------------------------------------
    Consensus = _extract(f'{REF}/models/tsm.py', 'SegmentConsensus', kind=ast.ClassDef)
    out = {}
    for i, (b, t, c) in enumerate([(1, 8, 12), (4, 8, 12), (3, 16, 5), (2, 1, 7)]):
        x = torch.from_numpy(rng.standard_normal((b, t, c)).astype(np.float32))
        y = Consensus('avg', 1)(x).squeeze(1)
        out[f'x{i}'] = x.numpy()
        out[f'y{i}'] = y.numpy()
    np.savez_compressed(f'{HERE}/ref_consensus.npz', **out)

    # ---- obo_mae / to_softmax -----------------------------------------------------------
    ref_obo = _extract(f'{REF}/utils/eval.py', 'obo_mae')
    ref_sm = _extract(f'{REF}/utils/visualize.py', 'to_softmax')
    obo_cases = []
    for n in (1, 2, 7, 50, 217):
        p = [int(v) for v in rng.integers(0, 40, n)]
        g = [int(v) for v in np.clip(np.array(p) + rng.integers(-3, 4, n), 0, None)]
        mae, obo = ref_obo(p, g)
        obo_cases.append(dict(preds=p, targets=g, mae=float(mae), obo=float(obo)))
    sm_cases = []
    for scale in (0.1, 1.0, 5.0, 30.0):
        for _ in range(4):
            logits = (rng.standard_normal(12) * scale).astype(np.float32)
            d = {str(i): float(v) for i, v in enumerate(logits)}
            o = ref_sm(d)
            sm_cases.append(dict(scores=d, softmax={k: float(v) for k, v in o.items()}))
    json.dump(dict(obo_mae=obo_cases, to_softmax=sm_cases), open(f'{HERE}/ref_metrics.json', 'w'))
    print('metrics cases:', len(obo_cases), len(sm_cases))


    analyze_count_vectors(rng)
    ckpt_remap_vectors()


def analyze_count_vectors(rng):
    """Execute the reference's analyze_count on synthetic evaluation CSVs.  pandas 2 removed DataFrame.append (the
    reference was written against pandas 1.x); the HARNESS supplies it as concat, the function body is untouched."""
    import contextlib
    import io
    import tempfile

    import pandas as pd
    ref_obo = _extract(f'{REF}/utils/eval.py', 'obo_mae')
    tree = ast.parse(open(f'{REF}/utils/eval.py').read())
    (node,) = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'analyze_count']
    ns = dict(pd=pd, np=np, obo_mae=ref_obo, Dict=typing.Dict, Optional=typing.Optional)
    exec(compile(ast.Module(body=[node], type_ignores=[]), 'eval.py', 'exec'), ns)
    ref_analyze = ns['analyze_count']
    if not hasattr(pd.DataFrame, 'append'):
        pd.DataFrame.append = lambda self, other, ignore_index=False: pd.concat([self, other], ignore_index=ignore_index)
    acts = ['situp', 'push_up', 'pull_up', 'jump_jack', 'squat', 'front_raise']
    cases = []
    for ci, (n_per, n_act, splits) in enumerate([(3, 2, ['test']), (5, 6, ['train', 'val', 'test']),
                                                  (1, 1, ['val']), (17, 4, ['test', 'val'])]):
        rows = []
        for sp in splits:
            for a in acts[:n_act]:
                for i in range(n_per + (ci == 1 and a == 'squat')):           # unequal group sizes in one case
                    gt = int(rng.integers(0, 40))
                    pred = max(0, gt + int(rng.integers(-3, 4)))
                    rows.append([f'{sp}_{a}_{i}.mp4', gt, pred, '[]', '[]', sp, a])
        order = rng.permutation(len(rows))                                     # first-appearance order matters
        df = pd.DataFrame([rows[i] for i in order],
                          columns=['name', 'gt_count', 'pred_count', 'gt_rep', 'pred_rep', 'split', 'action'])
        with tempfile.TemporaryDirectory() as tmp:
            src, dst = os.path.join(tmp, 'in.csv'), os.path.join(tmp, 'out.csv')
            df.to_csv(src)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ref_analyze(src, dst)
            cases.append(dict(in_csv=open(src).read(), out_csv=open(dst).read(), stdout=buf.getvalue()))
    json.dump(dict(source='workoutdetector/utils/eval.py:58-114 executed via ast (DataFrame.append supplied by the '
                          'harness as pd.concat)', pandas=pd.__version__, cases=cases),
              open(f'{HERE}/ref_analyze_count.json', 'w'))
    print('analyze_count cases:', len(cases))


def ckpt_remap_vectors():
    """Execute the remap statements of create_model (everything between torch.load and load_state_dict) on key lists
    shaped like the checkpoints the reference loads: official TSM .pth (module.*, new_fc), Lightning .ckpt (model.*)."""
    tree = ast.parse(open(f'{REF}/models/tsm.py').read())
    (fn,) = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'create_model']
    (branch,) = [n for n in fn.body if isinstance(n, ast.If) and 'checkpoint' in ast.dump(n.test)]
    stmts = [n for n in branch.body
             if not ('torch' in ast.dump(n) and 'load' in ast.dump(n)) and 'load_state_dict' not in ast.dump(n)]
    code = compile(ast.Module(body=stmts, type_ignores=[]), 'tsm.py', 'exec')
    trunk = ['conv1.weight', 'bn1.weight', 'bn1.bias', 'bn1.running_mean', 'bn1.running_var', 'bn1.num_batches_tracked',
             'layer1.0.conv1.net.weight', 'layer1.0.bn1.weight', 'layer1.0.conv2.weight', 'layer1.0.downsample.0.weight',
             'layer1.0.downsample.1.running_var', 'layer4.2.conv3.weight', 'layer4.2.bn3.bias']
    cases = []
    for name, prefix, fc, rows, num_class in [
            ('official TSM .pth, classifier rows == num_class', 'module.base_model.', 'module.new_fc', 12, 12),
            ('official TSM .pth (SSv2, 174 classes) loaded for 12-class finetuning', 'module.base_model.', 'module.new_fc', 174, 12),
            ('Lightning .ckpt', 'model.base_model.', 'model.new_fc', 12, 12),
            ('Lightning .ckpt, fc already named fc', 'model.base_model.', 'model.fc', 12, 12),
            ('DataParallel checkpoint whose classifier is already module.fc', 'module.base_model.', 'module.fc', 12, 12),
            ('two-class factory default', 'module.base_model.', 'module.new_fc', 2, 2)]:
        keys = [prefix + k for k in trunk] + [fc + '.weight', fc + '.bias']
        sd = OrderedDict((k, torch.zeros(rows if k.endswith(fc + '.weight') else 1, 3) + i) for i, k in enumerate(keys))
        src_of = {id(v): k for k, v in sd.items()}
        ns = dict(ckpt=dict(state_dict=sd), num_class=num_class, OrderedDict=OrderedDict)
        exec(code, ns)
        cases.append(dict(name=name, keys=keys, fc_rows=rows, num_class=num_class,
                          remapped=[[k, src_of[id(v)]] for k, v in ns['base_dict'].items()]))
    json.dump(dict(source='workoutdetector/models/tsm.py:451-473 (statements between torch.load and load_state_dict) '
                          'executed via ast', cases=cases), open(f'{HERE}/ref_ckpt_remap.json', 'w'), indent=1)
    print('checkpoint remap cases:', len(cases))


if __name__ == '__main__':
    main()
