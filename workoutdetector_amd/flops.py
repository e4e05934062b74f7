"""Algorithmic work of the TSM-R50 forward: per-layer GEMM shapes and MACs.

Used by ``bench.py`` (roofline accounting) and ``tools/``; derived from ``weights.conv_specs()`` and the spatial
schedule of ResNet-50 v1.5 (7x7 s2 stem, 3x3 s2 max-pool, stride on the 3x3 of each stage's first block).
Shift, BN fold, ReLU, pooling and the segment mean count zero FLOPs.
"""
from __future__ import annotations

from typing import Dict, List

from .weights import conv_specs


def _out(size: int, k: int, stride: int) -> int:
    return (size + 2 * (k // 2) - k) // stride + 1


def layer_table(height: int = 224, width: int = 224) -> List[Dict[str, int]]:
    """One row per conv launch-able layer: name, cin, cout, k, s (stride), m (output pixels per frame), macs per frame."""
    rows: List[Dict[str, int]] = []
    h, w = height, width
    block_in = None          # spatial size at the input of the current bottleneck
    for wkey, _bn, cout, cin, k in conv_specs():
        name = wkey[len('base_model.'):].replace('.net.weight', '').replace('.0.weight', '').replace('.weight', '')
        if name == 'conv1':
            ho, wo = _out(h, 7, 2), _out(w, 7, 2)
            rows.append(dict(name=name, cin=cin, cout=cout, k=k, s=2, m=ho * wo, macs=ho * wo * cout * cin * k * k))
            h, w = _out(ho, 3, 2), _out(wo, 3, 2)          # max-pool
            continue
        layer, block, part = name.split('.')
        stride = 2 if (block == '0' and layer != 'layer1') else 1
        if part == 'conv1':
            block_in = (h, w)
            s, ho, wo = 1, h, w
        elif part == 'conv2':
            s, ho, wo = stride, _out(h, 3, stride), _out(w, 3, stride)
            h, w = ho, wo
        elif part == 'conv3':
            s, ho, wo = 1, h, w
        else:                                              # downsample: 1x1 strided on the block input
            s, ho, wo = stride, _out(block_in[0], 1, stride), _out(block_in[1], 1, stride)
        rows.append(dict(name=name, cin=cin, cout=cout, k=k, s=s, m=ho * wo, macs=ho * wo * cout * cin * k * k))
    return rows


def macs_per_frame(height: int = 224, width: int = 224, num_class: int = 12) -> int:
    return sum(r['macs'] for r in layer_table(height, width)) + 2048 * num_class


def flops_per_clip(num_segments: int = 8, height: int = 224, width: int = 224, num_class: int = 12) -> float:
    return 2.0 * macs_per_frame(height, width, num_class) * num_segments
