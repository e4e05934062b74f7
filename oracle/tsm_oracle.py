"""torch-CPU fp32 restatement of the reference's TSM-ResNet50 eval forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: the network is
a plain dict of tensors keyed like the reference's ``TSM.state_dict()``:

    base_model.conv1.weight, base_model.bn1.{weight,bias,running_mean,running_var}
    base_model.layer{1..4}.{b}.conv1.net.weight      (conv1 is wrapped by TemporalShift,
                                                      workoutdetector/models/tsm.py:134-136)
    base_model.layer{L}.{b}.conv{2,3}.weight, .bn{1,2,3}.*
    base_model.layer{L}.0.downsample.0.weight, .downsample.1.*
    fc.weight, fc.bias

What each function follows in /root/reference (read as text, never imported):
  temporal_shift      workoutdetector/models/tsm.py:35-50
  shift placement     workoutdetector/models/tsm.py:125-137 ('blockres', n_round=1 for R50)
  resnet50 trunk      torchvision 0.13.0 ResNet/Bottleneck (v1.5: stride on conv2, bias-free
                      convs, BN eps 1e-5, maxpool k3 s2 p1), kept by tsm.py:250-251,264-281
  head                workoutdetector/models/tsm.py:409-419 (+ SegmentConsensus :165-174)
  tsm_forward_bf16    the same forward (tsm.py:409-419) with bf16 storage of weights / activations and fp32
                      accumulation written out: the oracle of BASELINE config 5 ("bf16 weights")
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
R50_BLOCKS = (3, 4, 6, 3)
R50_PLANES = (64, 128, 256, 512)
EXPANSION = 4


def temporal_shift(x: torch.Tensor, n_segment: int, fold_div: int = 8) -> torch.Tensor:
    """Zero-padded out-of-place shift over the segment axis (tsm.py:35-50).

    x: [N*T, C, H, W].  fold = C // fold_div.  Channels [0,fold) take frame t+1
    (zero at t = T-1), [fold,2fold) take frame t-1 (zero at t = 0), the rest copy.
    """
    nt, c, h, w = x.shape
    assert nt % n_segment == 0
    v = x.reshape(nt // n_segment, n_segment, c, h, w)
    fold = c // fold_div
    out = torch.zeros_like(v)
    out[:, :-1, :fold] = v[:, 1:, :fold]
    out[:, 1:, fold:2 * fold] = v[:, :-1, fold:2 * fold]
    out[:, :, 2 * fold:] = v[:, :, 2 * fold:]
    return out.reshape(nt, c, h, w)


def _bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], training=False, eps=BN_EPS)


def _conv1_key(sd: Dict[str, torch.Tensor], prefix: str) -> str:
    k = prefix + '.conv1.net.weight'
    return k if k in sd else prefix + '.conv1.weight'


def bottleneck(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, stride: int,
               n_segment: int, shift_div: int, is_shift: bool = True) -> torch.Tensor:
    """One torchvision Bottleneck with TSM's conv1 wrapped by the temporal shift."""
    identity = x
    h = temporal_shift(x, n_segment, shift_div) if is_shift else x
    h = F.conv2d(h, sd[_conv1_key(sd, prefix)])
    h = F.relu(_bn(h, sd, prefix + '.bn1'))
    h = F.conv2d(h, sd[prefix + '.conv2.weight'], stride=stride, padding=1)
    h = F.relu(_bn(h, sd, prefix + '.bn2'))
    h = F.conv2d(h, sd[prefix + '.conv3.weight'])
    h = _bn(h, sd, prefix + '.bn3')
    if prefix + '.downsample.0.weight' in sd:
        identity = F.conv2d(x, sd[prefix + '.downsample.0.weight'], stride=stride)
        identity = _bn(identity, sd, prefix + '.downsample.1')
    return F.relu(h + identity)


def stem(x: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    h = F.conv2d(x, sd['base_model.conv1.weight'], stride=2, padding=3)
    h = F.relu(_bn(h, sd, 'base_model.bn1'))
    return F.max_pool2d(h, kernel_size=3, stride=2, padding=1)


def trunk(x: torch.Tensor, sd: Dict[str, torch.Tensor], n_segment: int, shift_div: int = 8,
          is_shift: bool = True, taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """conv1..layer4 on [N*T,3,H,W] -> [N*T,2048,H/32,W/32]."""
    h = stem(x, sd)
    if taps is not None:
        taps['stem'] = h
    for li, nblocks in enumerate(R50_BLOCKS, start=1):
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            h = bottleneck(h, sd, f'base_model.layer{li}.{b}', stride, n_segment, shift_div, is_shift)
            if taps is not None:
                taps[f'layer{li}.{b}'] = h
    return h


def head(feat: torch.Tensor, sd: Dict[str, torch.Tensor], n_segment: int) -> torch.Tensor:
    """avgpool -> (dropout: eval no-op) -> fc per frame -> mean over segments (tsm.py:411-419)."""
    o = F.adaptive_avg_pool2d(feat, 1).flatten(1)
    o = F.linear(o, sd['fc.weight'], sd['fc.bias'])
    o = o.reshape(-1, n_segment, o.shape[-1])
    return o.mean(dim=1, keepdim=True).squeeze(1)


@torch.no_grad()
def tsm_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, n_segment: int = 8,
                shift_div: int = 8, is_shift: bool = True,
                taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """x: [B*T,3,H,W] or [B,T,3,H,W] fp32 -> raw logits [B,num_class] (before_softmax=True)."""
    if x.dim() == 5:
        x = x.reshape((-1,) + tuple(x.shape[2:]))
    assert x.dim() == 4 and x.shape[1] == 3 and x.shape[0] % n_segment == 0
    x = x.to(torch.float32)
    feat = trunk(x, sd, n_segment, shift_div, is_shift, taps)
    out = head(feat, sd, n_segment)
    if taps is not None:
        taps['logits'] = out
    return out


def conv_bn_act(x: torch.Tensor, w: torch.Tensor, bn: Tuple[torch.Tensor, ...], stride: int,
                padding: int, relu: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv -> BN(eval) -> (+residual) -> (ReLU): the per-op oracle for the HIP conv kernel.

    bn = (gamma, beta, running_mean, running_var).
    """
    g, b, m, v = bn
    h = F.conv2d(x, w, stride=stride, padding=padding)
    h = F.batch_norm(h, m, v, g, b, training=False, eps=BN_EPS)
    if residual is not None:
        h = h + residual
    return F.relu(h) if relu else h


# ---- bf16-storage restatement (BASELINE config 5) ----------------------------------------------------------------
# The same graph (tsm.py:409-419) with the roundings of a bf16-storage deployment written out: BatchNorm folded into
# the conv in fp32 (scale = gamma / sqrt(var + eps), w * scale, bias = beta - mean * scale), the folded weights and
# EVERY stored activation rounded to bfloat16 (round to nearest even), every sum accumulated in fp32.  A block's
# conv3 and its downsample branch accumulate into ONE fp32 sum before the single rounding of the block output (that
# is what "store once" means for the first block of a stage); the classifier reads the stored bf16 features and
# works in fp32.  Against THIS oracle the bf16 engine's error is fp32 accumulation order plus the rare rounding
# boundary flip, not the 2^-9 per layer of the format itself -- so the bar can be two orders tighter than against
# the fp32 oracle, and a dropped K-tile or a wrong residual source no longer hides under the format's own error.
def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest-even bfloat16 -> fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


def fold_bn(w: torch.Tensor, bn: Tuple[torch.Tensor, ...]) -> Tuple[torch.Tensor, torch.Tensor]:
    """(w * gamma / sqrt(var + eps), beta - mean * gamma / sqrt(var + eps)), every operation a correctly rounded fp32
    one.  Done in numpy: torch's vectorised CPU ``sqrt`` is 1 ulp off for about 1 % of inputs, and one ulp of the
    folded weight is enough to flip its bf16 rounding (0.4 % of that weight) -- the fold has to be reproducible to
    the bit for a bf16-storage oracle to mean anything."""
    import numpy as np
    g, b, m, v = (t.detach().to(torch.float32).numpy() for t in bn)
    scale = (g / np.sqrt(v + np.float32(BN_EPS))).astype(np.float32)
    wf = w.detach().to(torch.float32).numpy() * scale[:, None, None, None]
    return torch.from_numpy(wf.astype(np.float32)), torch.from_numpy((b - m * scale).astype(np.float32))


def _sd_bn(sd: Dict[str, torch.Tensor], prefix: str) -> Tuple[torch.Tensor, ...]:
    return (sd[prefix + '.weight'], sd[prefix + '.bias'], sd[prefix + '.running_mean'], sd[prefix + '.running_var'])


def conv_bn_act_bf16(x: torch.Tensor, w: torch.Tensor, bn: Tuple[torch.Tensor, ...], stride: int, padding: int,
                     relu: bool, residual: Optional[torch.Tensor] = None, round_output: bool = False) -> torch.Tensor:
    """Per-op oracle of a bf16-storage conv: operands (input, folded weights, residual) rounded to bf16, fp32
    accumulate, fp32 bias; the result is left UNROUNDED by default so that a test can ask "is the kernel's bf16 output
    a correct rounding of this value" (|got - want| <= 2^-8 |want|) without tripping over boundary flips."""
    wf, bias = fold_bn(w, bn)
    h = F.conv2d(bf16_round(x), bf16_round(wf), stride=stride, padding=padding) + bias[None, :, None, None]
    if residual is not None:
        h = h + bf16_round(residual)
    h = F.relu(h) if relu else h
    return bf16_round(h) if round_output else h


def _bottleneck_bf16(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, stride: int, n_segment: int,
                     shift_div: int, is_shift: bool) -> torch.Tensor:
    h = temporal_shift(x, n_segment, shift_div) if is_shift else x
    h = conv_bn_act_bf16(h, sd[_conv1_key(sd, prefix)], _sd_bn(sd, prefix + '.bn1'), 1, 0, True, round_output=True)
    h = conv_bn_act_bf16(h, sd[prefix + '.conv2.weight'], _sd_bn(sd, prefix + '.bn2'), stride, 1, True, round_output=True)
    w3, b3 = fold_bn(sd[prefix + '.conv3.weight'], _sd_bn(sd, prefix + '.bn3'))
    out = F.conv2d(h, bf16_round(w3))
    if prefix + '.downsample.0.weight' in sd:
        wd, bd = fold_bn(sd[prefix + '.downsample.0.weight'], _sd_bn(sd, prefix + '.downsample.1'))
        out = out + F.conv2d(x, bf16_round(wd), stride=stride) + (b3 + bd)[None, :, None, None]
    else:
        out = out + b3[None, :, None, None] + x
    return bf16_round(F.relu(out))


@torch.no_grad()
def tsm_forward_bf16(sd: Dict[str, torch.Tensor], x: torch.Tensor, n_segment: int = 8, shift_div: int = 8,
                     is_shift: bool = True, taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """``tsm_forward`` with bf16 storage of weights and activations, fp32 accumulation (see the block comment above)."""
    if x.dim() == 5:
        x = x.reshape((-1,) + tuple(x.shape[2:]))
    assert x.dim() == 4 and x.shape[1] == 3 and x.shape[0] % n_segment == 0
    h = conv_bn_act_bf16(x.to(torch.float32), sd['base_model.conv1.weight'], _sd_bn(sd, 'base_model.bn1'), 2, 3, True,
                         round_output=True)
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps['stem'] = h
    for li, nblocks in enumerate(R50_BLOCKS, start=1):
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            h = _bottleneck_bf16(h, sd, f'base_model.layer{li}.{b}', stride, n_segment, shift_div, is_shift)
            if taps is not None:
                taps[f'layer{li}.{b}'] = h
    out = head(h, sd, n_segment)
    if taps is not None:
        taps['logits'] = out
    return out


def layer_table(height: int = 224, width: int = 224) -> List[dict]:
    """Per-frame GEMM shapes of the 53 convs + fc, used by bench/roofline checks."""
    rows: List[dict] = []
    h, w = (height + 6 - 7) // 2 + 1, (width + 6 - 7) // 2 + 1
    rows.append(dict(name='conv1', cin=3, cout=64, k=7, s=2, m=h * w, macs=h * w * 64 * 147))
    h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    cin = 64
    for li, (nb, planes) in enumerate(zip(R50_BLOCKS, R50_PLANES), start=1):
        for b in range(nb):
            s = 2 if (b == 0 and li > 1) else 1
            ho, wo = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
            p = f'layer{li}.{b}'
            rows.append(dict(name=p + '.conv1', cin=cin, cout=planes, k=1, s=1, m=h * w,
                             macs=h * w * planes * cin))
            rows.append(dict(name=p + '.conv2', cin=planes, cout=planes, k=3, s=s, m=ho * wo,
                             macs=ho * wo * planes * planes * 9))
            rows.append(dict(name=p + '.conv3', cin=planes, cout=planes * EXPANSION, k=1, s=1,
                             m=ho * wo, macs=ho * wo * planes * EXPANSION * planes))
            if b == 0:
                rows.append(dict(name=p + '.downsample', cin=cin, cout=planes * EXPANSION, k=1, s=s,
                                 m=ho * wo, macs=ho * wo * planes * EXPANSION * cin))
            cin = planes * EXPANSION
            h, w = ho, wo
    return rows


def macs_per_frame(height: int = 224, width: int = 224, num_class: int = 12) -> int:
    return sum(r['macs'] for r in layer_table(height, width)) + 2048 * num_class
