"""Build libtsm_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m workoutdetector_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the working tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile
import sys
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), 'include')
# (TSM_LIB_PATH / TSM_BUILD_DEFS: tooling hooks for A/B builds of one kernel variant against another, e.g.
#  TSM_LIB_PATH=.../libtsm_hip_v1.so TSM_BUILD_DEFS='-DTSM_256_SCHED=1' python -m workoutdetector_amd.build --force)
LIB_PATH = os.environ.get('TSM_LIB_PATH') or os.path.join(PKG_DIR, 'libtsm_hip.so')
# One translation unit per kernel family (csrc/tsm_device.h lists them): objects are rebuilt only when their own source or
# a header changed, in parallel, then linked.
SOURCES = ['tsm_igemm.hip', 'tsm_bf16_256.hip', 'tsm_ws.hip', 'tsm_bneck.hip', 'tsm_conv31.hip', 'tsm_front.hip', 'tsm_fused23.hip',
           'tsm_stem.hip', 'tsm_ops.hip', 'tsm_engine.hip']
OBJ_DIR = os.path.join(CSRC, 'obj')
# Per-file compiler options.  tsm_bneck.hip: MFMA results in architectural VGPRs (the "vgprcd" form).  Its kernels pin 208
# weight registers in the accumulation half of the file, and with AGPRs in use the compiler otherwise puts EVERY MFMA
# result there too -- 168 v_accvgpr_read per step and wave (18 % of the step's vector-ALU instructions, on a kernel that is
# bound by exactly those) just to hand accumulators to the epilogues; the option leaves the weights where they are pinned.
# tsm_ws.hip: the same for the weight-stationary kernels (884 -> 16 v_accvgpr_read in the file; their epilogues run from
# registers).
EXTRA_FLAGS = {'tsm_bneck.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'], 'tsm_ws.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'],
               'tsm_front.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}
ARCH = 'gfx950'


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def _deps() -> List[str]:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))]
    deps.append(os.path.join(INCLUDE, 'tsm_hip.h'))
    return deps


def csrc_sha16() -> str:
    """First 16 hex digits of the sha256 over the kernel / engine sources (file names and contents, sorted): stamps
    measurements that depend on the exact kernel code (profiles/traffic.json), so that bench.py can tell a PMC figure
    measured on this source from a stale one."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            h.update(' '.join(EXTRA_FLAGS.get(name, [])).encode())      # per-file compiler options change the code object too
            with open(os.path.join(CSRC, name), 'rb') as f:
                h.update(f.read())
    h.update(b'include/tsm_hip.h')
    with open(os.path.join(INCLUDE, 'tsm_hip.h'), 'rb') as f:      # the ABI header is compiled into the engine object
        h.update(f.read())
    return h.hexdigest()[:16]


def build_defs() -> List[str]:
    """Extra compiler definitions of an A/B build (TSM_BUILD_DEFS); empty for the product."""
    return os.environ.get('TSM_BUILD_DEFS', '').split()


def build_id() -> str:
    """What the library built from THIS tree with THIS environment reports as tsm_build_id(): the sha of csrc/ (contents and
    per-file options) and, for an A/B build, of its extra definitions -- two variant libraries of one source never share a
    tune-cache line or pass for each other."""
    defs = build_defs()
    if not defs:
        return csrc_sha16()
    import hashlib
    return hashlib.sha256((csrc_sha16() + ' ' + ' '.join(defs)).encode()).hexdigest()[:16]


def library_build_id(path: str = None) -> str:
    """The id compiled into an existing library file, read from the FILE (the engine object carries the string
    'tsm-build-id:<id>'); '' when there is none.  Nothing is loaded or executed."""
    try:
        with open(path or LIB_PATH, 'rb') as f:
            blob = f.read()
    except OSError:
        return ''
    tag = b'tsm-build-id:'
    i = blob.find(tag)
    if i < 0:
        return ''
    j = blob.find(b'\0', i)
    return blob[i + len(tag):j].decode('ascii', 'replace')


def is_stale() -> bool:
    """A library is current when it carries the id of this tree (content, not mtimes: the git-ignored .so travels to the GPU
    box by copy, and a copy does not keep the order of modification times)."""
    return library_build_id() != build_id()


def _headers() -> List[str]:
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(INCLUDE, 'tsm_hip.h')]


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    defs = build_defs()
    # A/B builds (TSM_LIB_PATH / TSM_BUILD_DEFS) keep their objects apart from the product's -- and out of the tree (a dozen
    # variants are 30 MB of objects that would travel with every snapshot of the repository)
    tag = '' if not (defs or os.environ.get('TSM_LIB_PATH')) else '_' + hashlib_tag(' '.join(defs) + LIB_PATH)
    obj_dir = OBJ_DIR if not tag else os.path.join(tempfile.gettempdir(), 'tsm_hip_obj' + tag)
    os.makedirs(obj_dir, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())
    base = [_hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC'] + defs + ['-I', INCLUDE]
    jobs = []
    for src in SOURCES:
        path, obj = os.path.join(CSRC, src), os.path.join(obj_dir, src.replace('.hip', '.o'))
        # the engine object carries the build id that keys the tune cache: rebuilt whenever anything changed
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), newest_header) \
            or src == 'tsm_engine.hip'
        if stale:
            cmd = base + EXTRA_FLAGS.get(src, []) + ([f'-DTSM_BUILD_ID="{build_id()}"'] if src == 'tsm_engine.hip' else []) + \
                ['-c', path, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failed = []
    for src, proc in jobs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            failed.append(f'{src} ({proc.returncode}):\n{out}\n{err}')
    if failed:
        raise RuntimeError('hipcc failed: ' + '\n'.join(failed))
    cmd = [_hipcc(), f'--offload-arch={ARCH}', '-fPIC', '-shared', '-o', LIB_PATH] + \
          [os.path.join(obj_dir, s.replace('.hip', '.o')) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f'link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}')
    return LIB_PATH


def hashlib_tag(text: str) -> str:
    import hashlib
    return hashlib.sha256(text.encode()).hexdigest()[:10]


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
