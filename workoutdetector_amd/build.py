"""Build libtsm_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m workoutdetector_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the working tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), 'include')
# (TSM_LIB_PATH / TSM_BUILD_DEFS: tooling hooks for A/B builds of one kernel variant against another, e.g.
#  TSM_LIB_PATH=.../libtsm_hip_v1.so TSM_BUILD_DEFS='-DTSM_256_SCHED=1' python -m workoutdetector_amd.build --force)
LIB_PATH = os.environ.get('TSM_LIB_PATH') or os.path.join(PKG_DIR, 'libtsm_hip.so')
SOURCES = ['tsm_kernels.hip', 'tsm_engine.hip']
ARCH = 'gfx950'


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def _deps() -> List[str]:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
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
            with open(os.path.join(CSRC, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-shared'] + os.environ.get('TSM_BUILD_DEFS', '').split() + \
          ['-I', INCLUDE, '-o', LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f'hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}')
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
