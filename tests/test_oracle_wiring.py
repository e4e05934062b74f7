"""The oracle's ResNet-50 wiring against an INDEPENDENT public implementation.

torchvision (the reference's backbone, models/tsm.py:268) is absent from this image, so the oracle's trunk is a
restatement of the public v1.5 definition.  HuggingFace ``transformers.ResNetModel`` (constructed offline from a
config, random weights) is a separately written ResNet-50 v1.5 (stride on the 3x3, 1x1 strided shortcut in the first
block of every stage, 7x7 s2 stem + 3x3 s2 max-pool): loading the SAME weights into both and comparing the 2048-channel
feature map pins stride placement, block order, shortcut and BN/ReLU positions of the oracle -- everything except the
temporal shift, which is pinned against the reference's own function (tests/golden/ref_temporal_shift.npz)."""
import numpy as np
import pytest
import torch

from oracle import tsm_oracle
from workoutdetector_amd.weights import make_state_dict, to_torch

transformers = pytest.importorskip('transformers')


def _load_into_hf(model, sd):
    hf = {}

    def put(conv_key, bn_prefix, dst):
        hf[dst + '.convolution.weight'] = sd[conv_key]
        for a, b in (('weight', 'weight'), ('bias', 'bias'), ('running_mean', 'running_mean'), ('running_var', 'running_var')):
            hf[dst + '.normalization.' + b] = sd[bn_prefix + '.' + a]

    put('base_model.conv1.weight', 'base_model.bn1', 'embedder.embedder')
    for s, nb in enumerate((3, 4, 6, 3)):
        for b in range(nb):
            src, dst = f'base_model.layer{s + 1}.{b}', f'encoder.stages.{s}.layers.{b}'
            if b == 0:
                put(src + '.downsample.0.weight', src + '.downsample.1', dst + '.shortcut')
            put(src + '.conv1.net.weight', src + '.bn1', dst + '.layer.0')
            put(src + '.conv2.weight', src + '.bn2', dst + '.layer.1')
            put(src + '.conv3.weight', src + '.bn3', dst + '.layer.2')
    missing, unexpected = model.load_state_dict(hf, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing), (missing[:5], unexpected[:5])


def test_oracle_trunk_equals_huggingface_resnet50_v15():
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                       layer_type='bottleneck', hidden_act='relu', downsample_in_first_stage=False,
                       downsample_in_bottleneck=False)
    model = ResNetModel(cfg).eval()
    sd = to_torch(make_state_dict(5, 12))
    _load_into_hf(model, sd)
    x = torch.randn(3, 3, 96, 80, generator=torch.Generator().manual_seed(0))
    taps = {}
    with torch.no_grad():
        want = model(x, output_hidden_states=True)
        got = tsm_oracle.trunk(x, sd, n_segment=1, is_shift=False, taps=taps)
    assert tuple(got.shape) == (3, 2048, 3, 3)
    scale = float(want.last_hidden_state.abs().max())
    assert float((got - want.last_hidden_state).abs().max()) <= 1e-5 * scale
    # stage outputs too: hidden_states = (embedder output, stage 1..4)
    hs = want.hidden_states
    for i, name in enumerate(['stem', 'layer1.2', 'layer2.3', 'layer3.5', 'layer4.2']):
        s = float(hs[i].abs().max())
        assert float((taps[name] - hs[i]).abs().max()) <= 1e-5 * s, name
    # and the v1.5 property itself: the stride of the first block of stages 2-4 sits on the 3x3 conv
    assert model.encoder.stages[1].layers[0].layer[1].convolution.stride == (2, 2)
    assert model.encoder.stages[1].layers[0].layer[0].convolution.stride == (1, 1)
