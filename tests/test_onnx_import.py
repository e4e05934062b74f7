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


def _folded_export(sd, down_first=False):
    """Eval-mode export with BatchNorm folded into anonymous Conv weight/bias initialisers, wired like the network
    (conv1 -> conv2 -> conv3, downsample from the block input, Add, Relu).  ``down_first`` emits the downsample Conv
    of a block BEFORE its conv1 (a legal topological order some exporters / optimisers produce)."""
    nodes, inits = [], []
    cur = 'input'
    pending = []

    def conv(i, spec, src):
        wkey, bnp, cout, cin, k = spec
        scale = sd[bnp + '.weight'] / np.sqrt(sd[bnp + '.running_var'] + np.float32(EPS))
        w = (sd[wkey] * scale[:, None, None, None]).astype(np.float32)
        b = (sd[bnp + '.bias'] - sd[bnp + '.running_mean'] * scale).astype(np.float32)
        inits.extend([(f'onnx::Conv_{500 + 2 * i}', w), (f'onnx::Conv_{501 + 2 * i}', b)])
        return ('Conv', [src, f'onnx::Conv_{500 + 2 * i}', f'onnx::Conv_{501 + 2 * i}'], [f'c{i}'], f'Conv_{i}')

    specs = conv_specs()
    nodes.append(conv(0, specs[0], cur))
    nodes.append(('Relu', ['c0'], ['r0'], 'Relu_0'))
    nodes.append(('MaxPool', ['r0'], ['p0'], 'MaxPool_0'))
    cur, i = 'p0', 1
    while i < len(specs):
        has_down = i + 3 < len(specs) and '.downsample.' in specs[i + 3][0]
        block = [conv(i, specs[i], cur), ('Relu', [f'c{i}'], [f'r{i}'], f'Relu_{i}'),
                 conv(i + 1, specs[i + 1], f'r{i}'), ('Relu', [f'c{i + 1}'], [f'r{i + 1}'], f'Relu_{i + 1}'),
                 conv(i + 2, specs[i + 2], f'r{i + 1}')]
        identity = cur
        if has_down:
            d = conv(i + 3, specs[i + 3], cur)
            block = [d] + block if down_first else block + [d]
            identity = f'c{i + 3}'
        block += [('Add', [f'c{i + 2}', identity], [f'a{i}'], f'Add_{i}'), ('Relu', [f'a{i}'], [f'o{i}'], f'Relu_o{i}')]
        nodes += block
        cur = f'o{i}'
        i += 4 if has_down else 3
    del pending
    inits += [('fc.weight', sd['fc.weight']), ('fc.bias', sd['fc.bias'])]
    nodes.append(('GlobalAveragePool', [cur], ['feat'], 'gap'))
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


def _assert_same_folded_weights(got, sd):
    assert set(got) == set(sd)
    for wkey, bnp, cout, cin, k in conv_specs():
        # folding the imported (identity-BN) tensors must reproduce folding the original ones
        s0 = sd[bnp + '.weight'] / np.sqrt(sd[bnp + '.running_var'] + np.float32(EPS))
        w0, b0 = sd[wkey] * s0[:, None, None, None], sd[bnp + '.bias'] - sd[bnp + '.running_mean'] * s0
        s1 = got[bnp + '.weight'] / np.sqrt(got[bnp + '.running_var'] + np.float32(EPS))
        w1, b1 = got[wkey] * s1[:, None, None, None], got[bnp + '.bias'] - got[bnp + '.running_mean'] * s1
        np.testing.assert_allclose(w1, w0, rtol=2e-6, atol=1e-8, err_msg=wkey)
        np.testing.assert_allclose(b1, b0, rtol=1e-5, atol=1e-6, err_msg=bnp)
    assert np.array_equal(got['fc.weight'], sd['fc.weight']) and np.array_equal(got['fc.bias'], sd['fc.bias'])


@pytest.mark.parametrize('down_first', [False, True])
def test_folded_export_maps_convs_by_connectivity(tmp_path, sd, down_first):
    """layer1.0.conv3 and layer1.0.downsample.0 are both (256,64,1,1): only their position in the dataflow graph tells
    them apart, so the file order of the nodes must not matter (an exporter that emits the downsample conv
    first must not swap them silently)."""
    nodes, inits = _folded_export(sd, down_first=down_first)
    path = str(tmp_path / 'folded.onnx')
    write_model(path, nodes, inits)
    _assert_same_folded_weights(load_onnx_state_dict(path, 12), sd)


@pytest.mark.parametrize('style', ['eval', 'training'])
def test_real_torch_export_is_imported(tmp_path, sd, style):
    """A file the repo's writer did NOT produce: ``torch.onnx.export(module, randn(1,8,3,224,224), path,
    opset_version=11)`` exactly like scripts/export_model.py:43-46, of an nn.Module with the reference's module tree
    under the Lightning ``model.`` prefix (tests/_torch_tsm.py).  'eval' = the reference's deployment export (Conv +
    BatchNorm fused, anonymous onnx::Conv_### initialisers, ~2 600 nodes of traced shift arithmetic between the convs);
    'training' = an export that keeps BatchNormalization nodes and state-dict names.  The imported weights must drive
    the oracle to the same logits as the original state dict."""
    import torch

    from oracle import tsm_oracle
    from tests._torch_tsm import LitWrapper, TorchTSM, export_onnx
    path = str(tmp_path / f'tsm_{style}.onnx')
    export_onnx(LitWrapper(TorchTSM(num_class=12).load_engine_state_dict(sd)), path, training=(style == 'training'))
    inits, nodes = parse_onnx(path)
    assert sum(n['op_type'] == 'Conv' for n in nodes) == 53
    assert any(k.startswith('onnx::Conv_') for k in inits) == (style == 'eval')
    got = load_onnx_state_dict(path, 12)
    if style == 'training':
        assert set(got) == set(sd) and all(np.array_equal(got[k], sd[k]) for k in sd)
    else:
        _assert_same_folded_weights(got, sd)
    x = torch.randn(1, 8, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    want = tsm_oracle.tsm_forward({k: torch.from_numpy(v) for k, v in sd.items()}, x)
    have = tsm_oracle.tsm_forward({k: torch.from_numpy(np.asarray(v)) for k, v in got.items()}, x)
    assert float((have - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_rejects_wrong_models(tmp_path, sd):
    nodes, inits = _folded_export(sd)
    write_model(str(tmp_path / 'short.onnx'), nodes[:-12], inits)
    with pytest.raises(ValueError, match='Conv nodes'):
        load_onnx_state_dict(str(tmp_path / 'short.onnx'), 12)
    # a graph whose convs are not wired like a ResNet-50 (here: every conv reads the graph input) is refused
    flat = [(op, (['input'] + ins[1:]) if op == 'Conv' else ins, outs, name) for op, ins, outs, name in nodes]
    write_model(str(tmp_path / 'flat.onnx'), flat, inits)
    with pytest.raises(ValueError, match='cannot map the graph'):
        load_onnx_state_dict(str(tmp_path / 'flat.onnx'), 12)
    write_model(str(tmp_path / 'ok.onnx'), nodes, inits)
    with pytest.raises(ValueError, match='classifier'):
        load_onnx_state_dict(str(tmp_path / 'ok.onnx'), 5)
    open(tmp_path / 'empty.onnx', 'wb').write(b'')
    with pytest.raises(ValueError, match='no graph'):
        load_onnx_state_dict(str(tmp_path / 'empty.onnx'), 12)
