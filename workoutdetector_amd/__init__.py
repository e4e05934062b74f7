"""MI355X-native TSM-ResNet50 clip inference + repetition counting (hot path of iucario/WorkoutDetector).

The arithmetic lives in ``libtsm_hip.so`` (hand-written HIP for gfx950, C ABI in include/tsm_hip.h);
this package is the host side: engine binding, weights, clip pipeline, counter, evaluation.
Importing the package does not load the library; constructing a ``TsmEngine`` does, and fails loudly
without it (no CPU fallback).
"""
from .counting import RepCounter, obo_mae, pred_to_count, scores_to_preds, to_softmax  # noqa: F401

__all__ = ['RepCounter', 'obo_mae', 'pred_to_count', 'scores_to_preds', 'to_softmax', 'TsmEngine', 'create_model']


def __getattr__(name):
    if name in ('TsmEngine', 'create_model'):
        from . import engine
        return getattr(engine, name)
    raise AttributeError(name)
