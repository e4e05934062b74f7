"""ONNX weight importer (no onnx package offline): parsed from the protobuf wire format (CPU)."""
import numpy as np
import pytest

from tests._onnx_writer import write_model
from workoutdetector_amd.onnx_import import load_onnx_state_dict, parse_onnx
from workoutdetector_amd.weights import conv_specs, make_state_dict

EPS = 1e-5


def _named_export(sd):
    """Export that keeps state-dict names (Lightning prefix 'model.'), BatchNormalization nodes unfused."""
    inits = [('model.' + k.replace('fc.', 'new_fc.') if k.startswith('fc.') else 'model.' + k, v) for k, v in sd.items()]
    inits.append(('model.base_model.bn1.num_batches_tracked', np.array([7], dtype=np.int64)))
    nodes = []
    for i, (wkey, bnp, *_r) in enumerate(conv_specs()):
        nodes.append(('Conv', [f'x{i}', 'model.' + wkey], [f'c{i}'], f'Conv_{i}'))
        nodes.append(('BatchNormalization', [f'c{i}'] + ['model.' + bnp + s for s in ('.weight', '.bias', '.running_mean', '.running_var')],
                      [f'b{i}'], f'BN_{i}'))
    nodes.append(('Gemm', ['feat', 'model.new_fc.weight', 'model.new_fc.bias'], ['out'], 'Gemm_0'))
    return nodes, inits


def _folded_export(sd):
    """Eval-mode export with BatchNorm folded into anonymous Conv weight/bias initialisers."""
    nodes, inits = [], []
    for i, (wkey, bnp, cout, cin, k) in enumerate(conv_specs()):
        scale = sd[bnp + '.weight'] / np.sqrt(sd[bnp + '.running_var'] + np.float32(EPS))
        w = (sd[wkey] * scale[:, None, None, None]).astype(np.float32)
        b = (sd[bnp + '.bias'] - sd[bnp + '.running_mean'] * scale).astype(np.float32)
        inits += [(f'onnx::Conv_{500 + 2 * i}', w), (f'onnx::Conv_{501 + 2 * i}', b)]
        nodes.append(('Conv', [f'x{i}', f'onnx::Conv_{500 + 2 * i}', f'onnx::Conv_{501 + 2 * i}'], [f'c{i}'], f'Conv_{i}'))
        nodes.append(('Relu', [f'c{i}'], [f'r{i}'], f'Relu_{i}'))
    inits += [('fc.weight', sd['fc.weight']), ('fc.bias', sd['fc.bias'])]
    nodes.append(('Gemm', ['feat', 'fc.weight', 'fc.bias'], ['out'], 'Gemm_0'))
    return nodes, inits


@pytest.fixture(scope='module')
def sd():
    return make_state_dict(3, 12)


@pytest.mark.parametrize('raw', [True, False])
def test_named_initialisers_round_trip(tmp_path, sd, raw):
    nodes, inits = _named_export(sd)
    path = str(tmp_path / 'named.onnx')
    write_model(path, nodes, inits, raw=raw)
    parsed, pnodes = parse_onnx(path)
    assert len(parsed) == len(inits) and len(pnodes) == len(nodes) and pnodes[0]['op_type'] == 'Conv'
    got = load_onnx_state_dict(path, 12)
    assert set(got) == set(sd)
    for k in sd:
        assert np.array_equal(got[k], sd[k]), k


def test_folded_export_maps_convs_in_graph_order(tmp_path, sd):
    nodes, inits = _folded_export(sd)
    path = str(tmp_path / 'folded.onnx')
    write_model(path, nodes, inits)
    got = load_onnx_state_dict(path, 12)
    assert set(got) == set(sd)
    for wkey, bnp, cout, cin, k in conv_specs():
        # folding the imported (identity-BN) tensors must reproduce folding the original ones
        s0 = sd[bnp + '.weight'] / np.sqrt(sd[bnp + '.running_var'] + np.float32(EPS))
        w0, b0 = sd[wkey] * s0[:, None, None, None], sd[bnp + '.bias'] - sd[bnp + '.running_mean'] * s0
        s1 = got[bnp + '.weight'] / np.sqrt(got[bnp + '.running_var'] + np.float32(EPS))
        w1, b1 = got[wkey] * s1[:, None, None, None], got[bnp + '.bias'] - got[bnp + '.running_mean'] * s1
        np.testing.assert_allclose(w1, w0, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(b1, b0, rtol=1e-6, atol=1e-7)
    assert np.array_equal(got['fc.weight'], sd['fc.weight'])


def test_rejects_wrong_models(tmp_path, sd):
    nodes, inits = _folded_export(sd)
    write_model(str(tmp_path / 'short.onnx'), nodes[:-5], inits)
    with pytest.raises(ValueError, match='Conv nodes'):
        load_onnx_state_dict(str(tmp_path / 'short.onnx'), 12)
    write_model(str(tmp_path / 'ok.onnx'), nodes, inits)
    with pytest.raises(ValueError, match='classifier'):
        load_onnx_state_dict(str(tmp_path / 'ok.onnx'), 5)
    open(tmp_path / 'empty.onnx', 'wb').write(b'')
    with pytest.raises(ValueError, match='no graph'):
        load_onnx_state_dict(str(tmp_path / 'empty.onnx'), 12)
