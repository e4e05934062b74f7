"""A CPU stand-in with the onnxruntime duck type, for host-pipeline tests that must not need a GPU.
NOT a fallback of the product: it computes a toy per-clip function, batch-independent by construction."""
import numpy as np


class _Arg:
    def __init__(self, name):
        self.name = name


class StubModel:
    num_class = 12

    def __init__(self, seed=0, gain=4.0):
        rng = np.random.default_rng(seed)
        self.w = rng.standard_normal((12, 8 * 3)) * gain
        self.calls = 0

    def get_inputs(self):
        return [_Arg('input')]

    def run(self, output_names, feed):
        (x,) = feed.values()
        assert x.ndim == 5 and x.shape[1:3] == (8, 3) and x.dtype == np.float32
        self.calls += 1
        feat = x.astype(np.float64).mean(axis=(3, 4)).reshape(x.shape[0], 24)       # per segment, per channel
        feat = (feat - feat.mean(axis=1, keepdims=True)) / (feat.std(axis=1, keepdims=True) + 1e-6)
        return [(feat @ self.w.T).astype(np.float32)]


def synthetic_video(seed, frames, h, w, period=24):
    """uint8 [F,H,W,3]: brightness oscillates with ``period`` so that clip statistics change over time."""
    rng = np.random.default_rng(seed)
    t = np.arange(frames)
    level = 110 + 90 * np.sin(2 * np.pi * t / period)
    base = rng.integers(0, 40, size=(1, h, w, 3))
    tint = np.stack([np.cos(2 * np.pi * t / period), np.sin(2 * np.pi * t / (period * 0.7)), np.ones_like(t, float)], 1)
    vid = base + level[:, None, None, None] + 25 * tint[:, None, None, :]
    return np.clip(vid, 0, 255).astype(np.uint8)
