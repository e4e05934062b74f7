"""ctypes binding of libtsm_hip.so (include/tsm_hip.h).  Fails loudly when the library is missing:
there is no CPU fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from .build import LIB_PATH, build_id

ABI_VERSION = 7

MEM_HOST, MEM_DEVICE = 0, 1
LAYOUT_NTCHW, LAYOUT_NTHWC, LAYOUT_NTHWC4, LAYOUT_NTHWC8S, LAYOUT_NTHWC8B = 0, 1, 2, 3, 4
PIXEL_U8, PIXEL_F32 = 0, 1
DTYPE_F32, DTYPE_BF16X3, DTYPE_BF16 = 0, 1, 2
DTYPES = {'f32': DTYPE_F32, 'bf16x3': DTYPE_BF16X3, 'bf16': DTYPE_BF16}

STATUS_NAMES = {0: 'TSM_OK', -1: 'TSM_ERR_INVALID_ARG', -2: 'TSM_ERR_HIP', -3: 'TSM_ERR_NOT_FINALIZED',
                -4: 'TSM_ERR_MISSING_TENSOR', -5: 'TSM_ERR_SHAPE', -6: 'TSM_ERR_CAPACITY',
                -7: 'TSM_ERR_UNSUPPORTED'}


class TsmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('struct_size', 'num_class', 'num_segments', 'height', 'width',
                                         'shift_div', 'is_shift', 'max_clips', 'device_id', 'dtype')]


class TsmError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f'{STATUS_NAMES.get(status, status)}: {message}')
        self.status = status


EXPORTS = ('tsm_abi_version', 'tsm_build_id', 'tsm_trace_launches', 'tsm_launch_trace', 'tsm_create', 'tsm_destroy', 'tsm_last_error', 'tsm_set_tensor', 'tsm_finalize',
           'tsm_forward', 'tsm_tune', 'tsm_forward_tap', 'tsm_last_forward_ms', 'tsm_set_layer_timing', 'tsm_layer_times', 'tsm_conv_tiles', 'tsm_temporal_shift', 'tsm_conv_bn_act',
           'tsm_maxpool3x3s2', 'tsm_head', 'tsm_preprocess', 'tsm_gather_clips', 'tsm_scores_to_states')

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library and declare every prototype of include/tsm_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} is missing: build it with `python -m workoutdetector_amd.build` '
                          '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes
    # in one process cannot both own the device, so make sure torch's copy is the one already mapped
    # when libtsm_hip.so's DT_NEEDED is resolved: import torch first (it is the plumbing for device
    # memory and streams anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, fp, i32, i64 = C.c_void_p, C.c_void_p, C.c_int32, C.c_int64  # float* travel as raw addresses
    lib.tsm_abi_version.restype = C.c_int
    lib.tsm_abi_version.argtypes = []
    lib.tsm_build_id.restype = C.c_char_p
    lib.tsm_build_id.argtypes = []
    lib.tsm_trace_launches.restype = C.c_int
    lib.tsm_trace_launches.argtypes = [i32]
    lib.tsm_launch_trace.restype = i64
    lib.tsm_launch_trace.argtypes = [C.c_char_p, i64]
    lib.tsm_create.restype = C.c_int
    lib.tsm_create.argtypes = [C.POINTER(TsmConfig), C.POINTER(vp)]
    lib.tsm_destroy.restype = None
    lib.tsm_destroy.argtypes = [vp]
    lib.tsm_last_error.restype = C.c_char_p
    lib.tsm_last_error.argtypes = [vp]
    lib.tsm_set_tensor.restype = C.c_int
    lib.tsm_set_tensor.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    lib.tsm_finalize.restype = C.c_int
    lib.tsm_finalize.argtypes = [vp]
    lib.tsm_forward.restype = C.c_int
    lib.tsm_forward.argtypes = [vp, vp, i32, i32, i32, fp, vp]
    lib.tsm_forward_tap.restype = C.c_int
    lib.tsm_forward_tap.argtypes = [vp, vp, i32, i32, i32, C.c_char_p, fp, i64, C.POINTER(i64), vp]
    lib.tsm_last_forward_ms.restype = C.c_float
    lib.tsm_last_forward_ms.argtypes = [vp]
    lib.tsm_set_layer_timing.restype = C.c_int
    lib.tsm_set_layer_timing.argtypes = [vp, i32, i32]
    lib.tsm_layer_times.restype = C.c_int
    lib.tsm_layer_times.argtypes = [vp, i32, fp, i32, C.POINTER(i32)]
    lib.tsm_conv_tiles.restype = C.c_int
    lib.tsm_conv_tiles.argtypes = [vp, i32, C.POINTER(i32), i32, C.POINTER(i32)]
    lib.tsm_temporal_shift.restype = C.c_int
    lib.tsm_temporal_shift.argtypes = [fp, fp, i64, i32, i64, i32, i32, vp]
    lib.tsm_conv_bn_act.restype = C.c_int
    lib.tsm_conv_bn_act.argtypes = [fp] * 8 + [i32] * 11 + [vp]
    lib.tsm_maxpool3x3s2.restype = C.c_int
    lib.tsm_maxpool3x3s2.argtypes = [fp, fp, i32, i32, i32, i32, vp]
    lib.tsm_preprocess.restype = C.c_int
    lib.tsm_preprocess.argtypes = [vp, i32, i32, i32, i32, fp, i32, i32, i32, i32, vp]
    lib.tsm_tune.restype = C.c_int
    lib.tsm_tune.argtypes = [vp, i32, vp]
    lib.tsm_gather_clips.restype = C.c_int
    lib.tsm_gather_clips.argtypes = [vp, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, vp, vp]
    lib.tsm_head.restype = C.c_int
    lib.tsm_head.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, vp]
    lib.tsm_scores_to_states.restype = C.c_int
    lib.tsm_scores_to_states.argtypes = [fp, i32, i32, i32, C.c_float, vp, fp, vp]
    if lib.tsm_abi_version() != ABI_VERSION:
        raise ImportError(f'libtsm_hip.so ABI {lib.tsm_abi_version()} != binding {ABI_VERSION}; rebuild')
    # The library is git-ignored and travels prebuilt: refuse one that was not built from THIS tree (a stale .so would be
    # tested and benchmarked silently).  TSM_LIB_PATH names an A/B build on purpose and is taken as it is.
    have, want = lib.tsm_build_id().decode(), build_id()
    if have != want and not os.environ.get('TSM_LIB_PATH'):
        raise ImportError(f'{LIB_PATH} was built from other sources (build id {have}, this tree is {want}): '
                          'rebuild it with `python -m workoutdetector_amd.build`')
    _lib = lib
    return lib


def check(status: int, engine: Optional[int] = None) -> None:
    if status != 0:
        msg = load().tsm_last_error(engine)
        raise TsmError(status, msg.decode() if msg else '')
