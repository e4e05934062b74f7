"""Weights from an exported ``.onnx`` TSM model -> engine state dict, without the ``onnx`` package.

The reference deploys ``checkpoints/*.onnx`` produced by ``torch.onnx.export(model, sample[1,8,3,224,224],
opset_version=11)`` (workoutdetector/scripts/export_model.py:35-47, trainer.py:325-330) and runs it with
onnxruntime (utils/inference_count.py:620).  Neither ``onnx`` nor ``onnxruntime`` exists in this image, so
this module reads the protobuf wire format directly (ModelProto.graph -> initializer / node; only the
fields needed) and maps the tensors onto the engine's ``TSM.state_dict()`` keys:

  * initialisers that still carry state-dict names (``...base_model.layer1.0.conv1.net.weight``,
    BatchNormalization inputs ``...bn1.weight/bias/running_mean/running_var``): any prefix in front of
    ``base_model.`` / ``fc.`` / ``new_fc.`` is stripped;
  * exports where the exporter folded BatchNorm into the convolutions (eval-mode Conv+BN fusion: anonymous
    ``onnx::Conv_###`` weight + bias initialisers): every Conv node is identified by CONNECTIVITY, not by its
    position in the file -- the number of Conv nodes upstream of it in the dataflow graph fixes its place in the
    network (conv1 and downsample of a block see the same upstream set, conv2 one more, conv3 two more; the two
    1x1 convs that share a count always differ in output channels), the weight shape must agree, and anything
    ambiguous or missing raises -- and every conv gets an identity BatchNorm carrying its bias.

Tested on files written by torch's own exporter (``torch.onnx.export(..., opset_version=11)`` of an nn.Module with
the reference's module tree, both export styles; tests/_torch_tsm.py) and on hand-written files
(``tests/_onnx_writer.py``) for the malformed cases.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from .weights import conv_specs

BN_EPS = 1e-5


# ---- protobuf wire format -------------------------------------------------------------------------------
def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf: memoryview):
    """Yield (field_number, wire_type, value) for one message; LEN values are memoryview slices."""
    pos, end = 0, len(buf)
    while pos < end:
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield field, wt, val


def _packed_varints(buf: memoryview) -> List[int]:
    out, pos = [], 0
    while pos < len(buf):
        v, pos = _varint(buf, pos)
        out.append(v)
    return out


def _parse_tensor(buf: memoryview) -> Tuple[str, np.ndarray]:
    """TensorProto: dims=1, data_type=2, float_data=4, int64_data=7, name=8, raw_data=9."""
    dims: List[int] = []
    dtype, name, raw, floats, ints = 0, '', None, [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            dims += _packed_varints(v) if wt == 2 else [v]
        elif f == 2:
            dtype = v
        elif f == 4:
            floats += list(struct.unpack(f'<{len(v) // 4}f', bytes(v))) if wt == 2 else [struct.unpack('<f', v)[0]]
        elif f == 7:
            ints += _packed_varints(v) if wt == 2 else [v]
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
    if dtype == 1:      # FLOAT
        arr = np.frombuffer(raw, dtype='<f4') if raw is not None else np.asarray(floats, dtype=np.float32)
    elif dtype == 7:    # INT64
        arr = np.frombuffer(raw, dtype='<i8') if raw is not None else np.asarray(ints, dtype=np.int64)
    else:
        return name, np.zeros(0, dtype=np.float32)      # other types are never weights of this model
    return name, arr.reshape(dims).copy()


def _parse_node(buf: memoryview) -> dict:
    """NodeProto: input=1, output=2, name=3, op_type=4 (attributes are not needed)."""
    node = dict(input=[], output=[], name='', op_type='')
    for f, _wt, v in _fields(buf):
        if f == 1:
            node['input'].append(bytes(v).decode())
        elif f == 2:
            node['output'].append(bytes(v).decode())
        elif f == 3:
            node['name'] = bytes(v).decode()
        elif f == 4:
            node['op_type'] = bytes(v).decode()
    return node


def parse_onnx(path: str) -> Tuple[Dict[str, np.ndarray], List[dict]]:
    """(initialisers by name, nodes in graph order) of an ONNX ModelProto file."""
    data = memoryview(open(path, 'rb').read())
    graph = None
    for f, wt, v in _fields(data):
        if f == 7 and wt == 2:      # ModelProto.graph
            graph = v
    if graph is None:
        raise ValueError(f'{path}: no graph in the ONNX file')
    inits: Dict[str, np.ndarray] = OrderedDict()
    nodes: List[dict] = []
    for f, wt, v in _fields(graph):
        if f == 1 and wt == 2:      # GraphProto.node
            nodes.append(_parse_node(v))
        elif f == 5 and wt == 2:    # GraphProto.initializer
            name, arr = _parse_tensor(v)
            inits[name] = arr
    return inits, nodes


# ---- mapping to the engine's state dict ---------------------------------------------------------------------
def _strip_prefix(name: str) -> str:
    for anchor in ('base_model.', 'new_fc.', 'fc.'):
        i = name.find(anchor)
        if i >= 0 and (i == 0 or name[i - 1] == '.'):
            return name[i:].replace('new_fc.', 'fc.')
    return name


def _resolve_convs(path: str, nodes: List[dict], inits: Dict[str, np.ndarray], specs) -> List[dict]:
    """The Conv node of every entry of ``conv_specs()``, found by dataflow position + weight shape."""
    convs = [i for i, n in enumerate(nodes) if n['op_type'] == 'Conv']
    if len(convs) != len(specs):
        raise ValueError(f'{path}: {len(convs)} Conv nodes, a TSM-ResNet50 has {len(specs)}')
    bit = {ni: 1 << k for k, ni in enumerate(convs)}
    upstream: Dict[str, int] = {}          # tensor name -> bitmask of the Conv nodes it depends on
    depth: Dict[int, int] = {}
    for ni, n in enumerate(nodes):         # ONNX graphs are topologically sorted
        m = 0
        for t in n['input']:
            m |= upstream.get(t, 0)
        if ni in bit:
            depth[ni] = bin(m).count('1')
            m |= bit[ni]
        for t in n['output']:
            upstream[t] = m
    by_key: Dict[Tuple[int, tuple], List[int]] = {}
    for ni in convs:
        w = inits.get(nodes[ni]['input'][1]) if len(nodes[ni]['input']) > 1 else None
        if w is None:
            raise ValueError(f'{path}: Conv "{nodes[ni]["name"]}" has no initialiser weight')
        by_key.setdefault((depth[ni], tuple(w.shape)), []).append(ni)
    out, before = [], 0                    # `before` = Conv nodes upstream of the current block's input
    for idx, (wkey, _bnp, cout, cin, k) in enumerate(specs):
        if idx == 0:
            want = 0
        else:
            role = wkey.rsplit('.', 2)[-2] if '.net.' not in wkey else 'conv1'   # conv2 | conv3 | 0 (downsample) | conv1
            want = before + {'conv1': 0, '0': 0, 'conv2': 1, 'conv3': 2}[role]
        cands = by_key.get((want, (cout, cin, k, k)), [])
        if len(cands) != 1:
            raise ValueError(f'{path}: {len(cands)} Conv nodes with {want} upstream convs and weight {(cout, cin, k, k)} '
                             f'for {wkey}; cannot map the graph onto TSM-ResNet50')
        out.append(nodes[cands[0]])
        nxt = specs[idx + 1][0] if idx + 1 < len(specs) else ''
        if idx == 0 or (nxt.endswith('.conv1.net.weight') or nxt == ''):
            before = idx + 1               # a block is complete: everything so far is upstream of the next one
    return out


def load_onnx_state_dict(path: str, num_class: int) -> 'OrderedDict[str, np.ndarray]':
    """Engine state dict (``TsmEngine.load_state_dict``) from a TSM-R50 ``.onnx`` export."""
    inits, nodes = parse_onnx(path)
    specs = conv_specs()
    named = OrderedDict((_strip_prefix(k), v) for k, v in inits.items())
    if all(w in named or w.replace('.conv1.net.', '.conv1.') in named for w, *_ in specs):
        sd = OrderedDict((k, v) for k, v in named.items()
                         if (k.startswith('base_model.') or k.startswith('fc.')) and v.dtype == np.float32)
    else:
        sd = OrderedDict()
        for (wkey, bnp, cout, cin, k), node in zip(specs, _resolve_convs(path, nodes, inits, specs)):
            w = inits[node['input'][1]]
            b = inits[node['input'][2]] if len(node['input']) > 2 else np.zeros(cout, np.float32)
            sd[wkey] = w.astype(np.float32)
            # identity BatchNorm carrying the folded bias: scale = 1/sqrt(var + eps) = 1, bias = beta
            sd[bnp + '.weight'] = np.ones(cout, np.float32)
            sd[bnp + '.bias'] = b.astype(np.float32)
            sd[bnp + '.running_mean'] = np.zeros(cout, np.float32)
            sd[bnp + '.running_var'] = np.full(cout, 1.0 - BN_EPS, np.float32)
        fc = [n for n in nodes if n['op_type'] in ('Gemm', 'MatMul')]
        if not fc:
            raise ValueError(f'{path}: no Gemm/MatMul node for the classifier')
        w = next(inits[i] for i in fc[-1]['input'] if i in inits and inits[i].ndim == 2)
        if w.shape == (2048, num_class):
            w = w.T
        sd['fc.weight'] = np.ascontiguousarray(w, dtype=np.float32)
        bias = [inits[i] for i in fc[-1]['input'] if i in inits and inits[i].ndim == 1]
        if not bias:        # MatMul followed by Add
            adds = [n for n in nodes if n['op_type'] == 'Add' and fc[-1]['output'][0] in n['input']]
            bias = [inits[i] for n in adds for i in n['input'] if i in inits and inits[i].ndim == 1]
        sd['fc.bias'] = bias[0].astype(np.float32) if bias else np.zeros(num_class, np.float32)
    if tuple(sd['fc.weight'].shape) != (num_class, 2048):
        raise ValueError(f'{path}: classifier is {tuple(sd["fc.weight"].shape)}, expected ({num_class}, 2048)')
    return sd
