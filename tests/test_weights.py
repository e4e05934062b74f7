import numpy as np
import torch

from workoutdetector_amd.weights import conv_specs, make_state_dict, remap_checkpoint_keys, required_keys


def test_state_dict_structure():
    sd = make_state_dict(0, 12)
    specs = conv_specs()
    assert len(specs) == 53
    assert sum(v.size for k, v in sd.items() if k.endswith('.weight') and v.ndim == 4) == 23454912   # conv params
    assert sd['base_model.layer1.0.conv1.net.weight'].shape == (64, 64, 1, 1)
    assert sd['base_model.layer4.0.downsample.0.weight'].shape == (2048, 1024, 1, 1)
    assert sd['fc.weight'].shape == (12, 2048)
    assert set(required_keys()) == set(sd)
    again = make_state_dict(0, 12)
    assert all(np.array_equal(sd[k], again[k]) for k in sd)
    assert not np.array_equal(sd['fc.weight'], make_state_dict(1, 12)['fc.weight'])


def test_remap_checkpoint_keys_like_create_model():
    """tsm.py:451-473: last two entries are the classifier -> fc.* iff rows == num_class; every key loses
    its first dotted component ('module.' / 'model.')."""
    ck = {'module.base_model.conv1.weight': torch.zeros(64, 3, 7, 7),
          'module.base_model.layer1.0.conv1.net.weight': torch.zeros(64, 64, 1, 1),
          'module.new_fc.weight': torch.ones(12, 2048), 'module.new_fc.bias': torch.ones(12)}
    out = remap_checkpoint_keys(ck, num_class=12)
    assert list(out)[:2] == ['base_model.conv1.weight', 'base_model.layer1.0.conv1.net.weight']
    assert 'fc.weight' in out and 'fc.bias' in out and 'new_fc.weight' not in out
    out = remap_checkpoint_keys(ck, num_class=174)       # pretrained head of another size is dropped
    assert 'fc.weight' not in out and 'new_fc.weight' not in out


def test_remap_reproduces_the_executed_reference_block(golden_dir):
    """tests/golden/ref_ckpt_remap.json: key lists pushed through the reference's own remap statements
    (models/tsm.py:451-473, AST-extracted and executed by make_reference_vectors.py).  Same keys, same order, same
    source tensor behind every key -- module./model. prefixes, new_fc -> fc, classifier of the wrong row count dropped,
    and the reference's delete-after-copy of a classifier already named module.fc."""
    import json
    ref = json.load(open(f'{golden_dir}/ref_ckpt_remap.json'))
    assert len(ref['cases']) >= 6
    for case in ref['cases']:
        fc = case['keys'][-2]
        sd = {k: torch.zeros(case['fc_rows'] if k == fc else 1, 3) + i for i, k in enumerate(case['keys'])}
        src_of = {id(v): k for k, v in sd.items()}
        out = remap_checkpoint_keys(sd, case['num_class'])
        assert [[k, src_of[id(v)]] for k, v in out.items()] == case['remapped'], case['name']
        assert list(sd) == case['keys']                   # the caller's dict is not modified


def _to_mmaction(sd):
    """Inverse mapping, written independently: engine keys -> mmaction2 ResNetTSM / TSMHead names."""
    out = {}
    for k, v in sd.items():
        if k.startswith('fc.'):
            out['cls_head.fc_cls.' + k[3:]] = v
            continue
        p = k.split('.')[1:]
        if p[0] == 'conv1':
            out['backbone.conv1.conv.' + '.'.join(p[1:])] = v
        elif p[0] == 'bn1':
            out['backbone.conv1.bn.' + '.'.join(p[1:])] = v
        elif p[2] == 'downsample':
            out[f'backbone.{p[0]}.{p[1]}.downsample.{"conv" if p[3] == "0" else "bn"}.' + '.'.join(p[4:])] = v
        elif p[2].startswith('conv'):
            out[f'backbone.{p[0]}.{p[1]}.{p[2]}.conv.' + '.'.join(p[3:])] = v
        else:                                            # bnN of a block lives inside convN's ConvModule
            out[f'backbone.{p[0]}.{p[1]}.conv{p[2][-1]}.bn.' + '.'.join(p[3:])] = v
    return out


def test_remap_mmaction_keys_round_trip():
    """mmaction2 checkpoints of the reference's --mmlab branch (backbone.* ConvModules, cls_head.fc_cls)."""
    from workoutdetector_amd.weights import is_mmaction_state_dict, remap_mmaction_keys
    sd = make_state_dict(3, 12)
    mm = _to_mmaction(sd)
    mm['backbone.conv1.bn.num_batches_tracked'] = np.zeros((), np.int64)
    assert 'backbone.layer2.0.conv1.conv.net.weight' in mm and 'backbone.layer2.0.downsample.bn.running_var' in mm
    assert is_mmaction_state_dict(mm) and not is_mmaction_state_dict(sd)
    back = remap_mmaction_keys(mm)
    tracked = back.pop('base_model.bn1.num_batches_tracked')
    assert tracked.shape == () and set(back) == set(sd)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
