"""A minimal ONNX ModelProto writer (protobuf wire format by hand) for importer tests: the image has no
``onnx`` package.  Writes only what ``workoutdetector_amd.onnx_import`` reads: graph.node (input, output,
name, op_type) and graph.initializer (dims, data_type, name, raw_data or float_data)."""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _len_field(field, payload):
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def _tensor(name, arr, raw=True):
    arr = np.ascontiguousarray(arr)
    msg = b''.join(_varint(1 << 3 | 0) + _varint(d) for d in arr.shape)      # dims, unpacked
    dtype = {np.dtype('float32'): 1, np.dtype('int64'): 7}[arr.dtype]
    msg += _varint(2 << 3 | 0) + _varint(dtype)
    msg += _len_field(8, name.encode())
    if raw or dtype != 1:
        msg += _len_field(9, arr.tobytes())
    else:
        msg += _len_field(4, struct.pack(f'<{arr.size}f', *arr.ravel()))     # packed float_data
    return msg


def _node(op_type, inputs, outputs, name):
    msg = b''.join(_len_field(1, i.encode()) for i in inputs)
    msg += b''.join(_len_field(2, o.encode()) for o in outputs)
    msg += _len_field(3, name.encode()) + _len_field(4, op_type.encode())
    return msg


def write_model(path, nodes, initializers, raw=True):
    """nodes: [(op_type, inputs, outputs, name)], initializers: [(name, ndarray)]."""
    graph = b''.join(_len_field(1, _node(*n)) for n in nodes)
    graph += _len_field(2, b'torch_jit')
    graph += b''.join(_len_field(5, _tensor(k, v, raw)) for k, v in initializers)
    model = _varint(1 << 3 | 0) + _varint(6)                 # ir_version
    model += _len_field(2, b'pytorch') + _len_field(7, graph)
    with open(path, 'wb') as f:
        f.write(model)
