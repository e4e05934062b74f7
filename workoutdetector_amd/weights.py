"""Weights for the TSM-R50 engine: seeded synthetic state dicts and checkpoint key remapping.

No trained weights exist offline, so benches and parity tests use
a deterministic, numerically non-trivial state dict keyed exactly like the reference's
``TSM.state_dict()`` (workoutdetector/models/tsm.py:250-262; conv1 of every Bottleneck is
wrapped by TemporalShift, hence ``...conv1.net.weight``, tsm.py:134-136).

``remap_checkpoint_keys`` mirrors ``create_model``'s loader (tsm.py:451-473): the last two
entries of the checkpoint are the classifier, they become ``fc.weight/bias`` iff their row
count equals ``num_class``; every key loses its first dotted component.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Mapping, Tuple

import numpy as np

R50_BLOCKS = (3, 4, 6, 3)
R50_PLANES = (64, 128, 256, 512)
EXPANSION = 4


def conv_specs() -> List[Tuple[str, str, int, int, int]]:
    """(conv weight key, bn prefix, cout, cin, k) for the 53 convs of TSM-R50, in forward order."""
    specs = [('base_model.conv1.weight', 'base_model.bn1', 64, 3, 7)]
    cin = 64
    for li, (nb, planes) in enumerate(zip(R50_BLOCKS, R50_PLANES), start=1):
        for b in range(nb):
            p = f'base_model.layer{li}.{b}'
            specs.append((p + '.conv1.net.weight', p + '.bn1', planes, cin, 1))
            specs.append((p + '.conv2.weight', p + '.bn2', planes, planes, 3))
            specs.append((p + '.conv3.weight', p + '.bn3', planes * EXPANSION, planes, 1))
            if b == 0:
                specs.append((p + '.downsample.0.weight', p + '.downsample.1',
                              planes * EXPANSION, cin, 1))
            cin = planes * EXPANSION
    return specs


def make_state_dict(seed: int = 0, num_class: int = 12) -> 'OrderedDict[str, np.ndarray]':
    """Deterministic fp32 state dict (numpy arrays, torch layouts: conv OIHW, fc [cls, 2048]).

    He-normal convs; BN statistics are all non-trivial so the fold is exercised; the last BN of
    each residual branch is damped so activations stay O(1) through 16 blocks; the classifier uses
    std 0.05 (the reference's init std 0.001, tsm.py:260-262, gives logits too flat to
    discriminate between clips).
    """
    rng = np.random.default_rng(seed)
    sd: 'OrderedDict[str, np.ndarray]' = OrderedDict()

    def f32(a):
        return np.ascontiguousarray(a, dtype=np.float32)

    for wkey, bnp, cout, cin, k in conv_specs():
        fan_in = cin * k * k
        sd[wkey] = f32(rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / fan_in))
        damp = 0.35 if bnp.endswith('.bn3') else 1.0
        sd[bnp + '.weight'] = f32(rng.uniform(0.8, 1.2, cout) * damp)
        sd[bnp + '.bias'] = f32(rng.standard_normal(cout) * 0.1)
        sd[bnp + '.running_mean'] = f32(rng.standard_normal(cout) * 0.1)
        sd[bnp + '.running_var'] = f32(rng.uniform(0.6, 1.4, cout))
    sd['fc.weight'] = f32(rng.standard_normal((num_class, 512 * EXPANSION)) * 0.05)
    sd['fc.bias'] = f32(rng.standard_normal(num_class) * 0.1)
    return sd


def remap_checkpoint_keys(state_dict: Mapping[str, object], num_class: int) -> 'OrderedDict[str, object]':
    """Checkpoint ``state_dict`` (``module.``/``model.``-prefixed) -> engine keys, exactly as create_model does it
    (models/tsm.py:451-473; pinned by tests/golden/ref_ckpt_remap.json, produced by executing those statements):
    the LAST TWO entries are the classifier; they become ``fc.weight`` / ``fc.bias`` iff the weight has ``num_class``
    rows and are dropped otherwise; every key loses its first dotted component.  Like the reference, a classifier
    that is already called ``module.fc`` is dropped by the delete that follows the copy -- the reference then keeps
    its random-init fc (``strict=False``); the engine reports the missing ``fc.weight`` instead of guessing."""
    items = OrderedDict(state_dict)
    keys = list(items.keys())
    fc_w, fc_b = keys[-2], keys[-1]
    if items[fc_w].shape[0] == num_class:
        items['module.fc.weight'] = items[fc_w]
        items['module.fc.bias'] = items[fc_b]
    del items[fc_w]
    del items[fc_b]
    return OrderedDict(('.'.join(k.split('.')[1:]), v) for k, v in items.items())


def is_mmaction_state_dict(state_dict: Mapping[str, object]) -> bool:
    """mmaction2 ``Recognizer2D`` checkpoints (the reference's ``--mmlab`` branch, utils/inference_count.py:516-519,
    configs/tsm_MultiActionRepCount_sthv2.py:5-22) name their parts ``backbone.*`` / ``cls_head.*``."""
    return any(k.startswith('backbone.') for k in state_dict) and any(k.startswith('cls_head.') for k in state_dict)


def remap_mmaction_keys(state_dict: Mapping[str, object]) -> 'OrderedDict[str, object]':
    """mmaction2 0.24 ``ResNetTSM`` + ``TSMHead`` keys -> engine keys.

    mmaction builds every conv as a ``ConvModule`` (``.conv`` + ``.bn``) and wraps ``conv1.conv`` of each block in
    ``TemporalShift`` (``.net``): ``backbone.layer1.0.conv1.conv.net.weight`` -> ``base_model.layer1.0.conv1.net.weight``,
    ``backbone.layer1.0.conv1.bn.*`` -> ``base_model.layer1.0.bn1.*``, ``downsample.conv/bn`` -> ``downsample.0/1``,
    stem ``backbone.conv1.conv/bn`` -> ``base_model.conv1`` / ``base_model.bn1``, ``cls_head.fc_cls`` -> ``fc``.
    Same graph as the reference's own TSM (pytorch-style ResNet-50: stride on the 3x3; blockres shift, shift_div 8)."""
    out: 'OrderedDict[str, object]' = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith('cls_head.fc_cls.'):
            out['fc.' + k[len('cls_head.fc_cls.'):]] = v
            continue
        if not k.startswith('backbone.'):
            continue                                    # e.g. optimizer-side buffers
        parts = k[len('backbone.'):].split('.')
        if parts[0] == 'conv1':                         # stem ConvModule
            tail = parts[2:]
            name = ['conv1'] + tail if parts[1] == 'conv' else ['bn1'] + tail
        elif parts[0].startswith('layer'):
            layer, blk, mod, kind, tail = parts[0], parts[1], parts[2], parts[3], parts[4:]
            if mod == 'downsample':
                name = [layer, blk, 'downsample', '0' if kind == 'conv' else '1'] + tail
            elif kind == 'conv':
                name = [layer, blk, mod] + tail         # keeps a leading 'net' for the shifted conv1
            else:
                name = [layer, blk, 'bn' + mod[-1]] + tail
        else:
            continue
        out['base_model.' + '.'.join(name)] = v
    return out


def required_keys(num_class_known: bool = True) -> Iterable[str]:
    for wkey, bnp, *_ in conv_specs():
        yield wkey
        for s in ('.weight', '.bias', '.running_mean', '.running_var'):
            yield bnp + s
    yield 'fc.weight'
    yield 'fc.bias'


def to_torch(sd: Mapping[str, np.ndarray]) -> Dict[str, 'object']:
    import torch
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
