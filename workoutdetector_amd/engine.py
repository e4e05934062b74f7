"""TsmEngine: Python host side of the MI355X TSM-R50 clip-inference engine.

Drop-in for the two duck types the reference's hot path is written against:

  * onnxruntime.InferenceSession style (workoutdetector/utils/inference_count.py:265,273-275,
    scripts/eval_classification.py:29,44):   ``model.get_inputs()[0].name`` and
    ``model.run(None, {name: float32[B,T,3,H,W]}) -> [float32[B,num_class]]`` (any B >= 1; the
    reference's export is fixed at B = 1).
  * nn.Module style (tests/test_models.py:26-28):   ``model(x[B*T,3,H,W]) -> [B,num_class]``
    with the factory ``create_model(num_class, num_segments, base_model, checkpoint, device, ...)``
    (workoutdetector/models/tsm.py:422-476).

All arithmetic happens in libtsm_hip.so (hand-written HIP for gfx950) through the C ABI in
include/tsm_hip.h.  No CPU fallback: construction raises if the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Mapping, Optional, Sequence

import numpy as np

from . import _lib
from .weights import is_mmaction_state_dict, make_state_dict, remap_checkpoint_keys, remap_mmaction_keys


@dataclass
class NodeArg:
    """Minimal stand-in for onnxruntime.NodeArg (only what the reference reads)."""
    name: str
    shape: list
    type: str = 'tensor(float)'


def _as_f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


class TsmEngine:
    INPUT_NAME = 'input'
    OUTPUT_NAME = 'output'

    def __init__(self, num_class: int = 12, num_segments: int = 8, height: int = 224, width: int = 224,
                 shift_div: int = 8, is_shift: bool = True, max_clips: int = 32, device: int = 0,
                 state_dict: Optional[Mapping[str, object]] = None, dtype: str = 'f32'):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.num_class, self.num_segments = int(num_class), int(num_segments)
        self.height, self.width = int(height), int(width)
        self.max_clips, self.device = int(max_clips), int(device)
        if dtype not in _lib.DTYPES:
            raise ValueError(f'dtype must be one of {sorted(_lib.DTYPES)}, got {dtype!r}')
        self.dtype = dtype
        # layout tsm_preprocess must write for this engine to consume frames in place
        self.packed_layout = {'f32': _lib.LAYOUT_NTHWC4, 'bf16x3': _lib.LAYOUT_NTHWC8S,
                              'bf16': _lib.LAYOUT_NTHWC8B}[dtype]
        cfg = _lib.TsmConfig(C.sizeof(_lib.TsmConfig), num_class, num_segments, height, width, shift_div,
                             1 if is_shift else 0, max_clips, device, _lib.DTYPES[dtype])
        _lib.check(self._lib.tsm_create(C.byref(cfg), C.byref(self._h)))
        self._finalized = False
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, object], strict: bool = False) -> 'TsmEngine':
        """Hand every tensor to the engine (it folds BatchNorm and packs), then finalize.
        Non-strict like the reference's ``load_state_dict(base_dict, strict=False)`` (tsm.py:473):
        unknown keys are skipped, ``num_batches_tracked`` is ignored; missing tensors raise."""
        for name, value in state_dict.items():
            if name.endswith('num_batches_tracked'):
                continue
            arr = _as_f32(value.detach().cpu().numpy() if hasattr(value, 'detach') else value)
            shape = (C.c_int64 * arr.ndim)(*arr.shape)
            rc = self._lib.tsm_set_tensor(self._h, name.encode(), arr.ctypes.data, shape, arr.ndim)
            if rc == -1 and not strict and b'unknown tensor name' in (self._lib.tsm_last_error(self._h) or b''):
                continue
            _lib.check(rc, self._h)
        _lib.check(self._lib.tsm_finalize(self._h), self._h)
        self._finalized = True
        return self

    # ---- onnxruntime.InferenceSession duck type ---------------------------------------------------
    def get_inputs(self) -> List[NodeArg]:
        return [NodeArg(self.INPUT_NAME, [None, self.num_segments, 3, self.height, self.width])]

    def get_outputs(self) -> List[NodeArg]:
        return [NodeArg(self.OUTPUT_NAME, [None, self.num_class])]

    def run(self, output_names: Optional[Sequence[str]], input_feed: Dict[str, np.ndarray]) -> List[np.ndarray]:
        if output_names is not None and list(output_names) != [self.OUTPUT_NAME]:
            raise ValueError(f'unknown outputs {output_names}; engine has [{self.OUTPUT_NAME!r}]')
        if set(input_feed) != {self.INPUT_NAME}:
            raise ValueError(f'feed must have exactly the key {self.INPUT_NAME!r}, got {sorted(input_feed)}')
        x = input_feed[self.INPUT_NAME]
        want = (self.num_segments, 3, self.height, self.width)
        if x.ndim != 5 or tuple(x.shape[1:]) != want:
            raise ValueError(f'input must be [B,{want[0]},3,{want[2]},{want[3]}], got {tuple(x.shape)}')
        return [self.forward_host(_as_f32(x))]

    # ---- nn.Module duck type ----------------------------------------------------------------------
    def __call__(self, x):
        """x: [B*T,3,H,W] (or [B,T,3,H,W]) torch tensor (cpu or on this engine's GPU) or ndarray;
        returns logits [B,num_class] of the same kind."""
        is_torch = hasattr(x, 'is_cuda')
        shape = tuple(x.shape)
        if len(shape) == 4:
            if shape[0] % self.num_segments or shape[1:] != (3, self.height, self.width):
                raise ValueError(f'input must be [B*{self.num_segments},3,{self.height},{self.width}], got {shape}')
            b = shape[0] // self.num_segments
        elif len(shape) == 5 and shape[1:] == (self.num_segments, 3, self.height, self.width):
            b = shape[0]
        else:
            raise ValueError(f'bad input shape {shape}')
        if is_torch and x.is_cuda:
            return self.forward_device(x.reshape(b, self.num_segments, 3, self.height, self.width))
        arr = _as_f32(x.detach().numpy() if is_torch else x)
        out = self.forward_host(arr.reshape(b, self.num_segments, 3, self.height, self.width))
        if is_torch:
            import torch
            return torch.from_numpy(out)
        return out

    def eval(self) -> 'TsmEngine':  # nn.Module API used by callers; inference-only engine
        return self

    def to(self, *_args, **_kw) -> 'TsmEngine':
        return self

    # ---- forwards ---------------------------------------------------------------------------------
    def _floats_per_clip(self, layout: int) -> int:
        """float32 slots one clip occupies in ``layout`` (the C ABI takes bare pointers: sizes are checked here)."""
        t, h, w = self.num_segments, self.height, self.width
        per = {_lib.LAYOUT_NTCHW: 3 * h * w, _lib.LAYOUT_NTHWC: 3 * h * w, _lib.LAYOUT_NTHWC4: 4 * h * w,
               _lib.LAYOUT_NTHWC8S: 8 * h * ((w + 1) // 2), _lib.LAYOUT_NTHWC8B: 4 * h * ((w + 1) // 2)}
        if layout not in per:
            raise ValueError(f'unknown layout {layout}')
        return t * per[layout]

    def _check_clips(self, n_elems: int, b: int, layout: int) -> None:
        if b <= 0 or n_elems != b * self._floats_per_clip(layout):
            raise ValueError(f'clips hold {n_elems} float32 values; layout {layout} needs {self._floats_per_clip(layout)} '
                             f'per clip x {b} clips (T={self.num_segments}, H={self.height}, W={self.width})')

    def forward_host(self, clips: np.ndarray, layout: int = _lib.LAYOUT_NTCHW) -> np.ndarray:
        """clips: float32 [B,T,3,H,W] (or [B,T,H,W,3] with LAYOUT_NTHWC) in host memory."""
        self._need_finalized()
        clips = _as_f32(clips)
        b = clips.shape[0]
        self._check_clips(clips.size, b, layout)
        out = np.empty((b, self.num_class), dtype=np.float32)
        for s in range(0, b, self.max_clips):
            chunk = np.ascontiguousarray(clips[s:s + self.max_clips])
            o = out[s:s + chunk.shape[0]]
            _lib.check(self._lib.tsm_forward(self._h, chunk.ctypes.data, _lib.MEM_HOST, layout, chunk.shape[0],
                                             o.ctypes.data, None), self._h)
        return out

    def forward_device(self, clips, out=None, layout: int = _lib.LAYOUT_NTCHW):
        """clips: contiguous float32 CUDA tensor [B,T,3,H,W] on this engine's device.  Enqueues on
        torch's current stream and returns a CUDA tensor [B,num_class] (no host sync)."""
        import torch
        self._need_finalized()
        if not (clips.is_cuda and clips.dtype == torch.float32):
            raise ValueError('forward_device needs a float32 CUDA tensor')
        if clips.device.index != self.device:
            raise ValueError(f'tensor on cuda:{clips.device.index}, engine on cuda:{self.device}')
        clips = clips.contiguous()
        b = clips.shape[0]
        self._check_clips(clips.numel(), b, layout)
        if out is None:
            out = torch.empty((b, self.num_class), dtype=torch.float32, device=clips.device)
        elif not (out.is_cuda and out.device == clips.device and out.dtype == torch.float32 and out.is_contiguous()
                  and tuple(out.shape) == (b, self.num_class)):
            raise ValueError(f'out must be a contiguous float32 [{b},{self.num_class}] tensor on {clips.device}')
        stream = torch.cuda.current_stream(clips.device).cuda_stream
        for s in range(0, b, self.max_clips):
            n = min(self.max_clips, b - s)
            _lib.check(self._lib.tsm_forward(self._h, clips[s:s + n].data_ptr(), _lib.MEM_DEVICE, layout, n,
                                             out[s:s + n].data_ptr(), stream), self._h)
        return out

    def warmup(self, batch_sizes: Optional[Sequence[int]] = None) -> 'TsmEngine':
        """Tune every power-of-two bucket of the clip count that will occur (default: 1, 2, 4, ... max_clips) now
        (``tsm_tune``: a few hundred ms each of timed launches on the engine's own zeroed input buffer, or nothing when
        the tune cache file already holds the bucket), not inside the first real request."""
        import torch
        self._need_finalized()
        if batch_sizes is None:
            batch_sizes, b = [], 1
            while b < self.max_clips:
                batch_sizes.append(b)
                b *= 2
            batch_sizes.append(self.max_clips)
        stream = torch.cuda.current_stream(torch.device('cuda', self.device)).cuda_stream
        for b in batch_sizes:
            b = max(1, min(int(b), self.max_clips))
            _lib.check(self._lib.tsm_tune(self._h, b, stream), self._h)
        return self

    def forward_tap(self, clips: np.ndarray, stage: str) -> np.ndarray:
        """Activation after ``stage`` as NHWC float32 ndarray (parity tests)."""
        self._need_finalized()
        clips = _as_f32(clips)
        self._check_clips(clips.size, clips.shape[0], _lib.LAYOUT_NTCHW)
        n = clips.shape[0] * self.num_segments
        h1, w1 = (self.height - 1) // 2 + 1, (self.width - 1) // 2 + 1          # stem conv output
        hp, wp = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1                            # after the max-pool = layer1's size
        cap = n * max(h1 * w1 * 64, hp * wp * 256, self.height * self.width * 8)
        buf = np.empty(cap, dtype=np.float32)
        shape = (C.c_int64 * 4)()
        _lib.check(self._lib.tsm_forward_tap(self._h, clips.ctypes.data, _lib.MEM_HOST, _lib.LAYOUT_NTCHW,
                                             clips.shape[0], stage.encode(), buf.ctypes.data, cap, shape, None),
                   self._h)
        dims = tuple(int(v) for v in shape)
        return buf[:int(np.prod(dims))].reshape(dims).copy()

    # ---- per-launch timing (bench.py roofline) -------------------------------------------------------
    def launch_names(self) -> List[str]:
        """Names of the kernel launches of one forward, in launch order (matches tsm_layer_times)."""
        names = ['pack_input', 'conv1', 'maxpool']
        for li, nb in enumerate((3, 4, 6, 3), start=1):
            for b in range(nb):
                p = f'layer{li}.{b}'
                names += ([p + '.downsample'] if b == 0 else []) + [p + '.conv1', p + '.conv2', p + '.conv3']
        return names + ['head']

    TILE_NAMES = {0: 'heuristic', 1: '128x128', 2: '128x64', 3: '64x64', 4: '32x32', 5: '128x128w8', 6: '256x256', 7: 'ws',
                  8: '256x256p'}

    @classmethod
    def tile_name(cls, code: int) -> str:
        """``tile + 256 * split``: '64x64/splitK' = one workgroup per (tile, K segment), combined in segment order
        (segmented fp32 layers at small batch); '64x64/tailK' = only the tiles of the last, partly filled round of resident
        workgroups run that way (ConvParams::ksplit = 2); '+conv3' on a block's conv2 = conv2 + conv3 + residual run as one fused
        kernel (the conv3 entry of that block is then unused); '+block' on a block's conv1 = the whole Bottleneck runs as one
        launch (bf16 layer1.1 / layer1.2; the conv2 / conv3 entries are then unused); '+conv1' on a block's conv3 = that
        launch also runs the NEXT block's shift + conv1 (bf16 layer2: conv31_fused_kernel; the next block's conv1 entry is then
        unused); '+conv2' on a block's conv1 = that launch also runs the block's stride-2 conv2 (bf16 layer2.0: front_s2_kernel; the
        conv2 entry is then unused)."""
        return (cls.TILE_NAMES[code & 15] + ('/splitK' if code & 0x100 else '') + ('/tailK' if code & 0x200 else '') + ('+conv3' if code & 0x400 else '') +
                ('+block' if code & 0x800 else '') + ('+conv1' if code & 0x1000 else '') + ('+conv2' if code & 0x2000 else ''))

    def conv_tiles(self, n_clips: int) -> Dict[str, str]:
        """Tile shape the autotuner chose per conv launch for an ``n_clips`` forward."""
        buf = (C.c_int32 * 64)()
        n = C.c_int32()
        _lib.check(self._lib.tsm_conv_tiles(self._h, n_clips, buf, 64, C.byref(n)), self._h)
        names = [k for k in self.launch_names() if k not in ('pack_input', 'maxpool', 'head')]
        assert n.value == len(names)
        return {k: self.tile_name(buf[i]) for i, k in enumerate(names)}

    def set_layer_timing(self, n_forwards: int, only_conv3x3: bool = False) -> None:
        _lib.check(self._lib.tsm_set_layer_timing(self._h, n_forwards, int(only_conv3x3)), self._h)

    def layer_times_ms(self, forward_index: int) -> Dict[str, float]:
        buf = (C.c_float * 80)()
        n = C.c_int32()
        _lib.check(self._lib.tsm_layer_times(self._h, forward_index, C.addressof(buf), 80, C.byref(n)), self._h)
        names = self.launch_names()
        assert n.value == len(names), (n.value, len(names))
        return {k: float(buf[i]) for i, k in enumerate(names)}

    @property
    def last_forward_ms(self) -> float:
        return float(self._lib.tsm_last_forward_ms(self._h))

    def _need_finalized(self) -> None:
        if not self._finalized:
            raise _lib.TsmError(-3, 'load_state_dict() has not been called')

    def close(self) -> None:
        if getattr(self, '_h', None) is not None and self._h.value:
            self._lib.tsm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_model(num_class: int = 2, num_segments: int = 8, base_model: str = 'resnet50',
                 checkpoint: Optional[str] = None, device: Optional[object] = None, fc_lr5: bool = True,
                 is_shift: bool = True, shift_div: int = 8, shift_place: str = 'blockres',
                 consensus_type: str = 'avg', img_feature_dim: int = 256, non_local: bool = False,
                 height: int = 224, width: int = 224, max_clips: int = 32, seed: int = 0, dtype: str = 'f32',
                 **kwargs) -> TsmEngine:
    """Counterpart of the reference factory (tsm.py:422-476) returning a ready TsmEngine.

    ``checkpoint`` is a ``torch.save``d dict with a ``state_dict`` entry (mmaction2 ``backbone.*``/``cls_head.*``
    checkpoints of the reference's --mmlab branch are recognised and mapped by ``weights.remap_mmaction_keys``); its keys are remapped like
    the reference does (strip the first dotted component, last two entries are the classifier).  A path
    ending in ``.onnx`` (the reference's exported model, scripts/export_model.py:35-47) is read by
    ``onnx_import.load_onnx_state_dict``.
    Without a checkpoint the reference starts from torchvision's ImageNet weights, which cannot be
    fetched offline: the engine then gets the seeded synthetic weights of ``weights.make_state_dict``.
    """
    if base_model != 'resnet50':
        raise NotImplementedError(f'{base_model}: the engine implements resnet50 only')
    assert consensus_type in ('avg',), 'the engine implements the avg consensus'
    assert shift_place == 'blockres', 'the engine implements blockres placement'
    if non_local:
        raise NotImplementedError('non_local')
    dev = 0
    if device is not None:
        s = str(device)
        if s == 'cpu':
            raise RuntimeError('TsmEngine has no CPU path; pass a CUDA/HIP device')
        dev = int(s.split(':')[1]) if ':' in s else 0
    if checkpoint is not None and str(checkpoint).endswith('.onnx'):
        from .onnx_import import load_onnx_state_dict        # the reference's deployed artefact
        sd = load_onnx_state_dict(checkpoint, num_class)
    elif checkpoint is not None:
        import torch
        ckpt = torch.load(checkpoint, map_location='cpu')
        raw = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
        # mmaction2 checkpoints (the reference's --mmlab branch) vs the reference's own TSM / Lightning ones
        sd = remap_mmaction_keys(raw) if is_mmaction_state_dict(raw) else remap_checkpoint_keys(raw, num_class)
    else:
        sd = make_state_dict(seed=seed, num_class=num_class)
    return TsmEngine(num_class=num_class, num_segments=num_segments, height=height, width=width,
                     shift_div=shift_div, is_shift=is_shift, max_clips=max_clips, device=dev, state_dict=sd,
                     dtype=dtype)


# ---- launch trace (tests): which kernels did the calls inside the block launch? ----------------------------
class launch_trace:
    """``with launch_trace() as tr: ...`` records one line per kernel launch the library makes from THIS thread inside the
    block (``tsm_trace_launches`` / ``tsm_launch_trace``); afterwards ``tr.kernels`` is the list of kernel names as the launch
    sites spell them (template arguments of an enclosing launcher template resolved in a trailing ``[BM = 64, ...]``) and
    ``tr.ran('bneck_ws_kernel')`` asks for a family by prefix.  The bitwise tests of forced kernel forms use it to assert
    that the kernel under test is the one that ran -- a silent fall-back would make 'bit-identical' trivially true."""

    def __init__(self):
        self.kernels: List[str] = []

    def __enter__(self) -> 'launch_trace':
        _lib.load().tsm_trace_launches(1)
        return self

    def __exit__(self, *exc) -> None:
        lib = _lib.load()
        need = int(lib.tsm_launch_trace(None, 0))
        buf = C.create_string_buffer(need)
        lib.tsm_launch_trace(buf, need)
        lib.tsm_trace_launches(0)
        self.kernels = [ln for ln in buf.value.decode().split('\n') if ln]

    def ran(self, prefix: str) -> bool:
        return any(k.startswith(prefix) for k in self.kernels)

    def count(self, prefix: str) -> int:
        return sum(k.startswith(prefix) for k in self.kernels)


# ---- per-op wrappers over the C ABI (device tensors), used by tests ----------------------------------
def _ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(t) -> int:
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda_f32(**tensors) -> None:
    """The C ABI takes bare device pointers: refuse anything that is not a float32 CUDA tensor up front."""
    import torch
    for name, t in tensors.items():
        if t is None:
            continue
        if not (hasattr(t, 'is_cuda') and t.is_cuda and t.dtype == torch.float32):
            raise ValueError(f'{name} must be a float32 CUDA tensor')


def temporal_shift_nhwc(x, n_segment: int, fold_div: int = 8):
    """x: CUDA float32 [N*T, H, W, C] (NHWC) -> shifted copy (tsm.py:35-50)."""
    import torch
    _need_cuda_f32(x=x)
    x = x.contiguous()
    n, h, w, c = x.shape
    if n_segment <= 0 or n % n_segment:
        raise ValueError(f'{n} frames are not a whole number of {n_segment}-frame clips')
    y = torch.empty_like(x)
    _lib.check(_lib.load().tsm_temporal_shift(x.data_ptr(), y.data_ptr(), n, n_segment, h * w, c, fold_div,
                                              _stream(x)))
    return y


def conv_bn_act_nhwc(x, w, gamma, beta, mean, var, stride: int = 1, relu: bool = True, residual=None,
                     shift_segments: int = 0, fold_div: int = 8, dtype: str = 'f32'):
    """x NHWC [n,h,w,cin], w OIHW; returns NHWC [n,ho,wo,cout]."""
    import torch
    _need_cuda_f32(x=x, w=w, gamma=gamma, beta=beta, mean=mean, var=var, residual=residual)
    x = x.contiguous()
    n, hi, wi, cin = x.shape
    cout, wcin, k, k2 = w.shape
    if wcin != cin or k != k2 or any(tuple(t.shape) != (cout,) for t in (gamma, beta, mean, var)):
        raise ValueError(f'w {tuple(w.shape)} / BatchNorm vectors do not match x {tuple(x.shape)}')
    pad = k // 2
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    if residual is not None and tuple(residual.shape) != tuple(y.shape):
        raise ValueError(f'residual {tuple(residual.shape)} must have the output shape {tuple(y.shape)}')
    res = None if residual is None else residual.contiguous()
    args = [t.contiguous() for t in (w, gamma, beta, mean, var)]
    _lib.check(_lib.load().tsm_conv_bn_act(x.data_ptr(), *[a.data_ptr() for a in args], _ptr(res), y.data_ptr(),
                                           n, hi, wi, cin, cout, k, stride, int(relu), shift_segments, fold_div,
                                           _lib.DTYPES[dtype], _stream(x)))
    return y


def preprocess_frames(frames, resize: int = 256, crop: int = 224, scale_255: bool = False, packed: bool = True,
                      layout: Optional[int] = None):
    """HIP test transform.  frames: CUDA uint8 or float32 [n,H,W,3] (decoder layout, values 0..255).
    Returns float32 [n,crop,crop,4] (``packed``: feed ``forward_device(..., layout=LAYOUT_NTHWC4)``),
    [n,3,crop,crop] (``packed=False``), or with ``layout=engine.packed_layout`` the packed format of that
    engine (LAYOUT_NTHWC8S for a bf16x3 engine: a float32-typed buffer [n,crop,ceil(crop/2),8] holding one
    split-bf16 group per pixel pair; LAYOUT_NTHWC8B: [n,crop,ceil(crop/2),4] float slots = 8 bf16 per pair)."""
    import torch
    frames = frames.contiguous()
    if frames.dtype == torch.uint8:
        pixel = _lib.PIXEL_U8
    elif frames.dtype == torch.float32:
        pixel = _lib.PIXEL_F32
    else:
        raise ValueError(f'frames must be uint8 or float32, got {frames.dtype}')
    n, h, w, c = frames.shape
    if c != 3 or not frames.is_cuda:
        raise ValueError('frames must be a CUDA tensor [n,H,W,3]')
    if layout is None:
        layout = _lib.LAYOUT_NTHWC4 if packed else _lib.LAYOUT_NTCHW
    pairs = (crop + 1) // 2     # the bf16 formats store pixel PAIRS: one 8-element group = 2 pixels x 4 channels
    shape = {_lib.LAYOUT_NTHWC4: (n, crop, crop, 4), _lib.LAYOUT_NTHWC8S: (n, crop, pairs, 8),
             _lib.LAYOUT_NTHWC8B: (n, crop, pairs, 4),     # 8 bf16 = 16 bytes = 4 float slots per pair
             _lib.LAYOUT_NTCHW: (n, 3, crop, crop)}[layout]
    out = torch.empty(shape, dtype=torch.float32, device=frames.device)
    _lib.check(_lib.load().tsm_preprocess(frames.data_ptr(), pixel, n, h, w, out.data_ptr(), layout, resize, crop,
                                          int(scale_255), _stream(frames)))
    return out


def gather_clips(frames, first_frame: int, total_frames: int, first_clip: int, n_clips: int, out=None,
                 n_segment: int = 8, clip_step: int = 8, clip_stride: int = 2, pad_frame: Optional[int] = None):
    """The clip windows of the dataset loop on the GPU (``tsm_gather_clips``; the reference's ``vid[i:i + 16:2]`` for
    ``i in range(0, len(vid), 8)``, tail zero-padded: utils/inference_count.py:411-414) over transformed frames.

    frames: CUDA tensor [n, ...] holding every ``clip_stride``-th frame of the video from source frame
    ``clip_stride * first_frame`` on (any ``preprocess_frames`` layout); ``pad_frame`` (default: the last frame of the
    buffer) stands in for positions past ``total_frames``.  Returns ``out`` [n_clips, n_segment, ...] (allocated when
    None; else any contiguous CUDA tensor of that many bytes, e.g. a slice of a persistent batch buffer)."""
    import torch
    if not frames.is_cuda or not frames.is_contiguous():
        raise ValueError('frames must be a contiguous CUDA tensor')
    n = int(frames.shape[0])
    frame_bytes = int(frames[0].numel()) * frames.element_size()
    if out is None:
        out = torch.empty((n_clips, n_segment) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    if (not out.is_cuda or not out.is_contiguous() or out.device != frames.device
            or out.numel() * out.element_size() != n_clips * n_segment * frame_bytes):
        raise ValueError('out must be a contiguous CUDA tensor of n_clips * n_segment frames on the frames\' device')
    # (ranges of any length: the library cuts them into launches of <= 65535 rows itself)
    _lib.check(_lib.load().tsm_gather_clips(frames.data_ptr(), n, frame_bytes, int(first_frame), int(total_frames),
                                            n - 1 if pad_frame is None else int(pad_frame), int(first_clip), n_clips,
                                            n_segment, clip_step, clip_stride, out.data_ptr(), _stream(frames)))
    return out


def scores_to_states(logits, threshold: float = 0.5, softmax: bool = True, return_top: bool = False):
    """K9 on the GPU: CUDA float32 logits [n, num_class] -> int32 states [n] (utils/eval.py:153-164: softmax, first
    arg-max, class id if its score >= threshold else -1) and optionally the winning score.  Enqueues on torch's
    current stream; no host sync."""
    import torch
    _need_cuda_f32(logits=logits)
    if logits.dim() != 2 or logits.shape[0] == 0:
        raise ValueError(f'logits must be [n >= 1, num_class], got {tuple(logits.shape)}')
    logits = logits.contiguous()
    n, c = logits.shape
    states = torch.empty(n, dtype=torch.int32, device=logits.device)
    top = torch.empty(n, dtype=torch.float32, device=logits.device) if return_top else None
    _lib.check(_lib.load().tsm_scores_to_states(logits.data_ptr(), n, c, int(softmax), float(threshold),
                                                states.data_ptr(), _ptr(top), _stream(logits)))
    return (states, top) if return_top else states


def maxpool3x3s2_nhwc(x):
    import torch
    _need_cuda_f32(x=x)
    x = x.contiguous()
    n, hi, wi, c = x.shape
    y = torch.empty((n, (hi - 1) // 2 + 1, (wi - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().tsm_maxpool3x3s2(x.data_ptr(), y.data_ptr(), n, hi, wi, c, _stream(x)))
    return y


def head_nhwc(feat, fc_w, fc_b, n_segment: int):
    import torch
    _need_cuda_f32(feat=feat, fc_w=fc_w, fc_b=fc_b)
    feat = feat.contiguous()
    n, h, w, c = feat.shape
    if n_segment <= 0 or n % n_segment or tuple(fc_w.shape)[1:] != (c,) or tuple(fc_b.shape) != (fc_w.shape[0],):
        raise ValueError(f'feat {tuple(feat.shape)}, fc_w {tuple(fc_w.shape)}, fc_b {tuple(fc_b.shape)}, T={n_segment} do not fit')
    b = n // n_segment
    out = torch.empty((b, fc_w.shape[0]), dtype=torch.float32, device=feat.device)
    _lib.check(_lib.load().tsm_head(feat.data_ptr(), fc_w.contiguous().data_ptr(), fc_b.contiguous().data_ptr(),
                                    out.data_ptr(), b, n_segment, h * w, c, fc_w.shape[0], _stream(feat)))
    return out
