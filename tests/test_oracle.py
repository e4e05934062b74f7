"""The CPU oracle against the reference's own vectors and against its committed golden logits (CPU)."""
import json

import numpy as np
import torch

from oracle import transform_oracle, tsm_oracle
from tests._util import assert_close, make_input
from workoutdetector_amd.weights import make_state_dict, to_torch


def test_temporal_shift_against_executed_reference(golden_dir):
    z = np.load(f'{golden_dir}/ref_temporal_shift.npz')
    assert len(z['meta']) == 8
    for i, (nb, t, c, h, w, div) in enumerate(z['meta']):
        got = tsm_oracle.temporal_shift(torch.from_numpy(z[f'x{i}']), int(t), int(div))
        assert torch.equal(got, torch.from_numpy(z[f'y{i}'])), i


def test_consensus_against_executed_reference(golden_dir):
    z = np.load(f'{golden_dir}/ref_consensus.npz')
    for i in range(4):
        x = torch.from_numpy(z[f'x{i}'])
        got = x.mean(dim=1, keepdim=True).squeeze(1)
        assert torch.equal(got, torch.from_numpy(z[f'y{i}']))
        # the oracle head with an identity classifier on 1x1 features is exactly this consensus
        b, t, c = x.shape
        sd = {'fc.weight': torch.eye(c), 'fc.bias': torch.zeros(c)}
        np.testing.assert_allclose(tsm_oracle.head(x.reshape(b * t, c, 1, 1), sd, t).numpy(), z[f'y{i}'],
                                   rtol=1e-6, atol=1e-7)


def test_layer_table_matches_survey():
    """4.0871 GMAC/frame at 224^2, 5.3383 at 256^2, 53 convs."""
    rows = tsm_oracle.layer_table(224, 224)
    assert len(rows) == 53
    assert sum(r['macs'] for r in rows) == 4087136256
    assert abs(sum(r['macs'] for r in tsm_oracle.layer_table(256, 256)) / 1e9 - 5.3383) < 1e-4
    assert 2 * tsm_oracle.macs_per_frame() * 8 / 1e9 == 65.394573312


def test_oracle_reproduces_committed_golden(golden_dir):
    """Guards the oracle (and the seeded weight generator) against drift; small cases only, for CPU time."""
    gold = json.load(open(f'{golden_dir}/tsm_r50_logits.json'))
    for name in ('b3_t8_64', 'b5_t4_32', 'b1_t16_96x128'):
        case = gold[name]
        b, t, _, h, w = case['shape']
        sd = to_torch(make_state_dict(case['weight_seed'], 12))
        taps = {}
        y = tsm_oracle.tsm_forward(sd, torch.from_numpy(make_input(case['input_seed'], b, t, h, w)), n_segment=t,
                                   taps=taps)
        # same machine class, same library: tight, but not bitwise (thread count may change the blocking)
        assert_close(y.numpy(), np.array(case['logits'], np.float32), rtol=1e-4, atol_scale=1e-5, what=name)
        for k, v in case['tap_abs_mean'].items():
            assert abs(float(taps[k].abs().mean()) - v) <= 1e-4 * abs(v) + 1e-7, (name, k)


def test_shift_only_touches_quarter_of_channels():
    x = torch.randn(16, 64, 3, 3)
    y = tsm_oracle.temporal_shift(x, 8, 8)
    assert torch.equal(y[:, 16:], x[:, 16:])
    v, w = x.view(2, 8, 64, 3, 3), y.view(2, 8, 64, 3, 3)
    assert torch.equal(w[:, :-1, :8], v[:, 1:, :8]) and torch.equal(w[:, 1:, 8:16], v[:, :-1, 8:16])
    assert float(w[:, -1, :8].abs().sum()) == 0.0 and float(w[:, 0, 8:16].abs().sum()) == 0.0


def test_transform_oracle_shapes_and_quirk():
    """torchvision-0.13 tensor semantics: short side -> 256, int() on the long side, round() crop offsets."""
    assert transform_oracle.resized_hw(360, 206) == (447, 256)
    assert transform_oracle.resized_hw(272, 480) == (256, 451)
    assert transform_oracle.crop_offsets(447, 256) == (112, 16)
    vid = torch.randint(0, 256, (20, 36, 50, 3), dtype=torch.uint8)
    clip = transform_oracle.make_clip(vid, 16)            # frames 16, 18 then zero padding
    assert clip.dtype == torch.float32 and tuple(clip.shape) == (8, 36, 50, 3)
    assert torch.equal(clip[:2], vid[16:20:2].float()) and float(clip[2:].abs().sum()) == 0.0
    x = transform_oracle.clip_to_input(clip)
    assert tuple(x.shape) == (1, 8, 3, 224, 224)
    # no /255: a mid-grey frame maps to (128 - mean) / std, far outside the usual +-2.5 range
    grey = torch.full((8, 40, 40, 3), 128.0)
    v = transform_oracle.clip_to_input(grey)[0, 0, :, 100, 100]
    np.testing.assert_allclose(v.numpy(), [(128 - m) / s for m, s in zip(transform_oracle.MEAN, transform_oracle.STD)],
                               rtol=1e-6)


def test_product_flop_accounting_equals_the_oracle_table():
    """bench.py prices its roofline with workoutdetector_amd.flops (the oracle is only its cpu_baseline leg): that
    table, derived from weights.conv_specs(), must equal the oracle's independently written one on every shape."""
    from workoutdetector_amd import flops
    for hw in [(224, 224), (256, 256), (96, 128), (225, 640), (64, 64)]:
        a = {r['name']: r for r in flops.layer_table(*hw)}
        b = {r['name']: r for r in tsm_oracle.layer_table(*hw)}
        assert a == b, hw
        assert flops.macs_per_frame(*hw, num_class=7) == tsm_oracle.macs_per_frame(*hw, num_class=7)
    assert flops.flops_per_clip(8, 224, 224, 12) / 1e9 == 65.394573312
