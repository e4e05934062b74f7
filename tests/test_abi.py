"""The C-ABI library builds, loads and exports exactly what include/tsm_hip.h declares (CPU, no compute)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from workoutdetector_amd import _lib
    from workoutdetector_amd.build import build_library
    build_library()
    return _lib.load()


def _declared():
    text = open(os.path.join(ROOT, 'include', 'tsm_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tsm_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported(lib):
    from workoutdetector_amd import _lib
    from workoutdetector_amd.build import LIB_PATH
    declared = _declared()
    assert declared == sorted(_lib.EXPORTS)
    out = subprocess.run(['nm', '-D', '--defined-only', LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r' T (tsm_[a-z0-9_]+)', out)))
    assert exported == declared
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_no_kernel_stub_is_missing_from_the_library():
    """A kernel whose HOST stub was dropped at compile time (seen with hipcc 7.2: a template-dependent constant inside the
    scalar-offset argument of a buffer builtin in a lambda -- no diagnostic, the object simply lacks the stub and its device
    code) leaves an undefined tsm:: symbol that only fails at the first launch: refuse it here, on the CPU."""
    from workoutdetector_amd.build import LIB_PATH, build_library
    build_library()
    out = subprocess.run(['nm', '-D', '--undefined-only', '-C', LIB_PATH], capture_output=True, text=True, check=True).stdout
    missing = [ln for ln in out.splitlines() if 'tsm::' in ln]
    assert not missing, missing


def test_abi_version_and_config_layout(lib):
    from workoutdetector_amd import _lib
    assert lib.tsm_abi_version() == _lib.ABI_VERSION == 7
    assert ctypes.sizeof(_lib.TsmConfig) == 40          # 10 x int32, matches struct tsm_config


def test_library_carries_the_build_id_of_this_tree(lib):
    """tsm_build_id() == build.build_id() (sha of csrc/ + the ABI header + per-file options [+ TSM_BUILD_DEFS]) == the tag found
    in the FILE without loading it: the three views a loader, a bench line and an incremental build use."""
    from workoutdetector_amd import build
    have = lib.tsm_build_id().decode()
    assert have == build.build_id() == build.library_build_id() and len(have) == 16 and int(have, 16) >= 0
    assert not build.is_stale()


def test_a_library_built_from_other_sources_is_refused(tmp_path):
    """The .so is git-ignored and travels prebuilt: _lib.load() must refuse one whose build id is not the tree's (it would be
    tested and benchmarked silently), and build.is_stale() must see it whatever the modification times say.  Done on a COPY
    of the library whose embedded id is patched, loaded in a child process (TSM_LIB_PATH would accept an A/B build on
    purpose, so the child points LIB_PATH at the copy by hand)."""
    import shutil
    import sys
    from workoutdetector_amd import build
    fake = str(tmp_path / 'libtsm_hip.so')
    shutil.copy(build.LIB_PATH, fake)
    blob = open(fake, 'rb').read()
    tag = b'tsm-build-id:' + build.build_id().encode()
    assert blob.count(tag) >= 1
    open(fake, 'wb').write(blob.replace(tag, b'tsm-build-id:' + b'0123456789abcdef'))
    os.utime(fake, None)                                    # newer than every source: the mtime rule would have kept it
    assert build.library_build_id(fake) == '0123456789abcdef' != build.build_id()
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from workoutdetector_amd import build, _lib\n'
            'build.LIB_PATH = _lib.LIB_PATH = %r\n'
            'assert build.is_stale()\n'
            'try:\n'
            '    _lib.load()\n'
            'except ImportError as e:\n'
            '    assert "built from other sources" in str(e) and "0123456789abcdef" in str(e), e\n'
            '    print("refused")\n') % (ROOT, fake)
    env = {k: v for k, v in os.environ.items() if k not in ('TSM_LIB_PATH', 'TSM_BUILD_DEFS')}
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and 'refused' in out.stdout, out.stderr


def test_ab_build_definitions_change_the_build_id(monkeypatch):
    """two variant libraries of one source (TSM_BUILD_DEFS) must not share tune-cache lines: the id hashes the
    definitions too."""
    from workoutdetector_amd import build
    plain = build.build_id()
    monkeypatch.setenv('TSM_BUILD_DEFS', '-DTSM_OUT_AUX=16')
    a = build.build_id()
    monkeypatch.setenv('TSM_BUILD_DEFS', '-DTSM_OUT_AUX=2')
    b = build.build_id()
    assert len({plain, a, b}) == 3 and plain == build.csrc_sha16()


def test_launch_trace_is_off_by_default_and_per_thread(lib):
    """tsm_trace_launches / tsm_launch_trace without a GPU: nothing launches, so the trace is empty; the size protocol
    (returns the bytes needed, copies when the buffer suffices) and the per-thread state are checked here, the contents
    on the GPU (tests/test_bf16_gpu.py, test_engine_gpu.py: "the kernel under test is the one that ran")."""
    import threading
    assert lib.tsm_launch_trace(None, 0) == 1                     # just the terminator
    assert lib.tsm_trace_launches(1) == 0
    buf = ctypes.create_string_buffer(4)
    assert lib.tsm_launch_trace(buf, 4) == 1 and buf.value == b''
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.tsm_launch_trace(None, 0)))
    t.start(); t.join()
    assert seen == [1]
    assert lib.tsm_trace_launches(0) == 0


def test_create_rejects_bad_config_before_touching_the_gpu(lib):
    from workoutdetector_amd import _lib
    h = ctypes.c_void_p()
    cfg = _lib.TsmConfig(4, 12, 8, 224, 224, 8, 1, 1, 0, 0)          # wrong struct_size
    assert lib.tsm_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and not h.value
    assert b'struct_size' in lib.tsm_last_error(None)
    cfg = _lib.TsmConfig(40, 12, 8, 224, 224, 7, 1, 1, 0, 0)         # shift_div 7 unsupported
    assert lib.tsm_create(ctypes.byref(cfg), ctypes.byref(h)) == -7
    cfg = _lib.TsmConfig(40, 0, 8, 224, 224, 8, 1, 1, 0, 0)
    assert lib.tsm_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg = _lib.TsmConfig(40, 12, 16, 1024, 1024, 8, 1, 1024, 0, 0)   # 2^32 output rows: beyond 32-bit row indices
    assert lib.tsm_create(ctypes.byref(cfg), ctypes.byref(h)) == -6 and b'2^31' in lib.tsm_last_error(None)
    lib.tsm_destroy(None)                                              # must be a no-op
    assert lib.tsm_forward(None, None, 0, 0, 1, None, None) == -1


def test_engine_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a GPU construction raises (on a GPU box this test is moot)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from workoutdetector_amd._lib import TsmError
    from workoutdetector_amd.engine import TsmEngine, create_model
    with pytest.raises(TsmError) as ei:
        TsmEngine(max_clips=1)
    assert ei.value.status == -2
    with pytest.raises(RuntimeError):
        create_model(num_class=12, device='cpu')


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under workoutdetector_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'workoutdetector_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f


def _build_c_smoke(tmp_path):
    from workoutdetector_amd.build import LIB_PATH, PKG_DIR
    exe = str(tmp_path / 'abi_c_smoke')
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'tests', 'abi_c_smoke.c'), '-o', exe, LIB_PATH, f'-Wl,-rpath,{PKG_DIR}',
           '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,/opt/rocm/lib']      # (hipMalloc / hipMemcpy for the device pipeline)
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_plain_c_and_links(lib, tmp_path):
    """include/tsm_hip.h compiles as C99 with -Wall -Werror and a C program links against the library; the
    error paths it exercises need no GPU."""
    exe = _build_c_smoke(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert 'error paths ok' in out.stdout
