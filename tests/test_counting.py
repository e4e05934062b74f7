"""Counter / metrics of the product against the reference's own vectors (CPU)."""
import json

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import counting_oracle
from workoutdetector_amd.counting import (RepCounter, eval_count, obo_mae, pred_to_count, scores_to_preds,
                                          softmax_rows, to_softmax)


def _kats(golden_dir):
    return json.load(open(f'{golden_dir}/counter_kats.json'))['kats']


def test_reference_known_answer_vectors(golden_dir):
    """tests/test_inference_count.py:8-48, the docstring example and the notebook trace of the reference."""
    kats = _kats(golden_dir)
    assert len(kats) == 8
    for k in kats:
        for impl in (pred_to_count, counting_oracle.pred_to_count):
            count, reps = impl(k['preds'], k['step'])
            assert count == k['count'], k['src']
            if k['reps'] is not None:
                assert reps == k['reps'], k['src']


def test_against_executed_reference(golden_dir):
    """144 seeded cases produced by executing the reference's own pred_to_count body."""
    ref = json.load(open(f'{golden_dir}/ref_pred_to_count.json'))
    assert len(ref['cases']) >= 100
    for c in ref['cases']:
        want = (c['count'], c['reps'])
        assert pred_to_count(c['preds'], c['step']) == want
        assert counting_oracle.pred_to_count(c['preds'], c['step']) == want


@given(st.lists(st.integers(-1, 11), max_size=80), st.sampled_from([1, 7, 8]))
@settings(max_examples=300, deadline=None)
def test_product_equals_oracle_and_invariants(preds, step):
    count, reps = pred_to_count(preds, step)
    assert (count, reps) == counting_oracle.pred_to_count(preds, step)
    assert len(reps) == 2 * count and all(r % step == 0 for r in reps)
    assert reps == sorted(reps) or count <= 1 or all(reps[i] <= reps[i + 1] for i in range(0, len(reps), 2))
    kept = [p for p in preds if p != -1]
    assert count <= sum(1 for a, b in zip(kept, kept[1:]) if b % 2 == 1 and a == b - 1)
    # streaming form: identical after every prefix
    rc = RepCounter(step)
    for i, p in enumerate(preds):
        rc.push(p)
        assert rc.count == pred_to_count(preds[:i + 1], step)[0]
    assert rc.reps == reps


def test_empty_and_background_only():
    assert pred_to_count([], 8) == (0, [])
    assert pred_to_count([-1] * 9, 8) == (0, [])
    assert pred_to_count([5], 8) == (0, [])


def test_softmax_and_metrics_against_executed_reference(golden_dir):
    m = json.load(open(f'{golden_dir}/ref_metrics.json'))
    for c in m['to_softmax']:
        got = to_softmax(c['scores'])
        assert list(got) == list(c['softmax'])
        np.testing.assert_allclose([got[k] for k in c['softmax']], list(c['softmax'].values()), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(counting_oracle.softmax(list(c['scores'].values())), list(c['softmax'].values()),
                                   rtol=1e-6, atol=1e-8)
    for c in m['obo_mae']:
        assert obo_mae(c['preds'], c['targets']) == (c['mae'], c['obo'])
        assert counting_oracle.obo_mae(c['preds'], c['targets']) == (c['mae'], c['obo'])


def test_scores_to_preds_threshold_and_ties():
    logits = np.array([[0.0] * 12, [5.0] + [0.0] * 11, [0.0] * 11 + [9.0], [2.0, 2.0] + [-90.0] * 10], np.float32)
    assert scores_to_preds(logits) == [-1, 0, 11, 0]            # flat -> below 0.5; tie -> first index (0.5 >= 0.5)
    assert scores_to_preds(logits, softmax=False, threshold=4.0) == [-1, 0, 11, -1]
    assert scores_to_preds(np.zeros((0, 12), np.float32)) == []
    assert scores_to_preds(logits) == counting_oracle.scores_to_preds(logits.tolist())
    p = softmax_rows(logits)
    np.testing.assert_allclose(p.sum(1), 1.0, rtol=1e-6)


def test_eval_count_property():
    """tests/test_repcount_dataset.py:66-85 of the reference: predictions = gt +- 1 => obo == 1 and
    mae == mean(1/gt)."""
    gt = {f'v{i}': c for i, c in enumerate([3, 8, 1, 12, 30, 5])}
    pred = {k: v + (1 if i % 2 else -1) for i, (k, v) in enumerate(gt.items())}
    mae, obo = eval_count(pred, gt)
    assert obo == 1.0
    assert mae == pytest.approx(np.mean([1 / v for v in gt.values()]))
    assert (mae, obo) == counting_oracle.eval_count(pred, gt)
    assert eval_count({'a': 4}, {'a': 0}) == (0.0, 0.0)


def test_analyze_count_is_the_reference_drop_in(golden_dir, tmp_path, capsys):
    """utils/eval.py:58-114 executed on synthetic evaluation CSVs (tests/golden/make_reference_vectors.py): same
    signature (csv, out_csv), same output CSV text (columns action,split,mae,obo_acc,total,avg_count; OBO as a count;
    per-split 'all' rows with the int(mae * n) re-accumulation) and the same printout."""
    import inspect
    import json

    from workoutdetector_amd import eval as tsm_eval
    assert list(inspect.signature(tsm_eval.analyze_count).parameters) == ['csv', 'out_csv']
    ref = json.load(open(f'{golden_dir}/ref_analyze_count.json'))
    assert len(ref['cases']) >= 4
    for i, case in enumerate(ref['cases']):
        src, dst = tmp_path / f'in{i}.csv', tmp_path / f'out{i}.csv'
        src.write_text(case['in_csv'])
        capsys.readouterr()
        assert tsm_eval.analyze_count(str(src), str(dst)) is None
        assert dst.read_text() == case['out_csv'], i
        assert capsys.readouterr().out == case['stdout'], i
    tsm_eval.analyze_count(str(tmp_path / 'in0.csv'), None)          # out_csv=None: print only
    # a (split, action) pair without videos divides by zero in the reference's obo_mae; same here
    import pandas as pd
    df = pd.read_csv(tmp_path / 'in1.csv', index_col=0)
    df = df[~((df.split == 'val') & (df.action == 'squat'))]
    df.to_csv(tmp_path / 'hole.csv')
    with pytest.raises(ZeroDivisionError):
        tsm_eval.analyze_count(str(tmp_path / 'hole.csv'), None)
