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
