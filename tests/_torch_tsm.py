"""An nn.Module spelling of the TSM-ResNet50 eval graph with the reference's module tree, for EXPORT tests only.

Test infrastructure, like oracle/: it exists so that ``torch.onnx.export`` (the exporter the reference uses,
workoutdetector/scripts/export_model.py:35-47, trainer.py:325-330) can write a REAL ``.onnx`` file for
``workoutdetector_amd.onnx_import`` to read -- a file this repository's own writer (tests/_onnx_writer.py) did not
produce.  Module names follow ``TSM.state_dict()``: ``base_model.{conv1,bn1,layerL.B.{conv1.net,bn1,conv2,bn2,conv3,bn3,
downsample.0,downsample.1}}`` and ``new_fc`` (workoutdetector/models/tsm.py:134-136,250-262); the Lightning wrapper adds
the ``model.`` prefix (trainer.py:25-40).  The forward flattens ``[B,T,3,H,W]`` to ``[B*T,3,H,W]`` first (the
``x.view(-1, 3, 224, 224)`` commented out in trainer.py:38-40, without which the 5-D export sample cannot run)."""
import torch
import torch.nn as nn

from oracle.tsm_oracle import EXPANSION, R50_BLOCKS, R50_PLANES, temporal_shift


class _Shifted(nn.Module):
    """TemporalShift wrapper: the wrapped conv is the attribute ``net`` (tsm.py:17-32)."""

    def __init__(self, net, n_segment, fold_div):
        super().__init__()
        self.net, self.n_segment, self.fold_div = net, n_segment, fold_div

    def forward(self, x):
        return self.net(temporal_shift(x, self.n_segment, self.fold_div))


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, down, n_segment, fold_div):
        super().__init__()
        self.conv1 = _Shifted(nn.Conv2d(cin, planes, 1, bias=False), n_segment, fold_div)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * EXPANSION, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * EXPANSION)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = nn.Sequential(nn.Conv2d(cin, planes * EXPANSION, 1, stride, bias=False),
                                        nn.BatchNorm2d(planes * EXPANSION)) if down else None

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        identity = x if self.downsample is None else self.downsample(x)
        return self.relu(out + identity)


class _Trunk(nn.Module):
    def __init__(self, n_segment, fold_div):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (nb, planes) in enumerate(zip(R50_BLOCKS, R50_PLANES), start=1):
            blocks = []
            for b in range(nb):
                blocks.append(_Bottleneck(cin, planes, 2 if (b == 0 and li > 1) else 1, b == 0, n_segment, fold_div))
                cin = planes * EXPANSION
            setattr(self, f'layer{li}', nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.avgpool(x).flatten(1)


class TorchTSM(nn.Module):
    def __init__(self, num_class=12, n_segment=8, fold_div=8):
        super().__init__()
        self.n_segment = n_segment
        self.base_model = _Trunk(n_segment, fold_div)
        self.new_fc = nn.Linear(512 * EXPANSION, num_class)

    def forward(self, x):
        x = x.view((-1,) + tuple(x.shape[-3:]))
        out = self.new_fc(self.base_model(x))
        out = out.view(-1, self.n_segment, out.shape[-1])
        return out.mean(dim=1, keepdim=True).squeeze(1)

    def load_engine_state_dict(self, sd):
        """Engine / oracle keys (``fc.*``) -> this module's (``new_fc.*``); strict."""
        self.load_state_dict({k.replace('fc.', 'new_fc.') if k.startswith('fc.') else k: torch.as_tensor(v)
                              for k, v in sd.items()}, strict=False)
        missing = [k for k in self.state_dict() if not k.endswith('num_batches_tracked')
                   and (k.replace('new_fc.', 'fc.') not in sd)]
        assert not missing, missing
        return self


class LitWrapper(nn.Module):
    """The reference exports its Lightning module, whose network is the attribute ``model`` (trainer.py:25-40)."""

    def __init__(self, net):
        super().__init__()
        self.model = net

    def forward(self, x):
        return self.model(x)


def export_onnx(module, path, sample_shape=(1, 8, 3, 224, 224), training=False):
    """``torch.onnx.export(model, sample, path, opset_version=11)`` as scripts/export_model.py:43-46 calls it, through
    torch's TorchScript exporter (``dynamo=False``).  The ``onnx`` Python package is absent from this image and the
    exporter imports it in ONE post-pass, ``_add_onnxscript_fn``, which re-serialises the model only when the graph
    holds onnxscript custom functions (never the case here: the pass returns its input bytes unchanged); the harness
    replaces that pass by the identity, everything else -- tracing, ONNX lowering, eval-mode Conv+BatchNorm fusion,
    protobuf serialisation -- is torch's own C++ exporter."""
    import warnings

    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils, utils
    saved = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            module = module.train() if training else module.eval()
            if training:        # tracing runs one forward: keep it from moving the BatchNorm running statistics
                for m in module.modules():
                    if isinstance(m, nn.BatchNorm2d):
                        m.momentum = 0.0
            mode = torch.onnx.TrainingMode.TRAINING if training else torch.onnx.TrainingMode.EVAL
            torch.onnx.export(module, torch.randn(*sample_shape), path, opset_version=11, dynamo=False, training=mode,
                              do_constant_folding=not training)
    finally:
        onnx_proto_utils._add_onnxscript_fn = saved
        assert utils is not None
    return path
