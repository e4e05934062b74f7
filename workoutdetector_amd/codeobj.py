"""Resource figures of the BUILT gfx950 code object: registers, LDS, scratch and the occupancy they allow.

``libtsm_hip.so`` carries its device code as clang offload bundles in ``.hip_fatbin`` (one per translation unit); a gfx950 entry is an ELF whose
``NT_AMDGPU_METADATA`` note is a msgpack document with one record per kernel (``.vgpr_count`` -- the unified register
count, accumulation registers included --, ``.agpr_count``, ``.sgpr_count``, ``.group_segment_fixed_size`` = static LDS,
``.private_segment_fixed_size`` = scratch, spill counts).  Several kernels of this library sit right at a budget (the
fused weight-stationary 3x3: 510 of 512 registers, 162 048 of 163 840 B of LDS; the segmented 64x64 fp32 tile: 96
registers for 5 waves per SIMD), so the figures DESIGN.md quotes are asserted on the built binary by
``tests/test_code_objects.py`` -- a compiler bump that starts spilling in a K loop fails a CPU test, not a benchmark.

    python -m workoutdetector_amd.codeobj            # table of every kernel
"""
from __future__ import annotations

import struct
import subprocess
from typing import Dict, List, Optional

SIMD_VGPRS = 512          # unified VGPR + AGPR file per SIMD lane (gfx90a / gfx942 / gfx950)
VGPR_GRANULE = 8
LDS_PER_CU = 160 * 1024   # gfx950
SIMDS_PER_CU = 4
MAX_WAVES_PER_SIMD = 8


def _gfx950_elfs(lib_path: str) -> List[bytes]:
    """The gfx950 code object of every offload bundle in the library: one bundle per translation unit (csrc/tsm_*.hip)."""
    data = open(lib_path, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    elfs, base = [], data.find(magic)
    if base < 0:
        raise ValueError(f'{lib_path}: no uncompressed clang offload bundle (.hip_fatbin)')
    while base >= 0:
        n, = struct.unpack_from('<Q', data, base + len(magic))
        off = base + len(magic) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from('<QQQ', data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if 'gfx950' in triple and size:
                elfs.append(data[base + o:base + o + size])
        base = data.find(magic, base + len(magic))
    if not elfs:
        raise ValueError(f'{lib_path}: no gfx950 code object in the bundles')
    return elfs


def kernel_metadata(lib_path: Optional[str] = None) -> Dict[str, dict]:
    """{demangled kernel name: metadata record (+ 'waves_per_simd', 'workgroups_per_cu' for its static resources)}."""
    import msgpack
    if lib_path is None:
        from .build import LIB_PATH as lib_path      # noqa: N811
    records: List[dict] = []
    for elf in _gfx950_elfs(lib_path):
        shoff, = struct.unpack_from('<Q', elf, 0x28)
        shentsize, shnum, _ = struct.unpack_from('<HHH', elf, 0x3A)
        for k in range(shnum):
            sh = struct.unpack_from('<IIQQQQIIQQ', elf, shoff + k * shentsize)
            if sh[1] != 7:                               # SHT_NOTE
                continue
            pos, end = sh[4], sh[4] + sh[5]
            while pos < end:
                namesz, descsz, typ = struct.unpack_from('<III', elf, pos)
                pos += 12 + ((namesz + 3) & ~3)
                desc = elf[pos:pos + descsz]
                pos += (descsz + 3) & ~3
                if typ == 32:                            # NT_AMDGPU_METADATA
                    records += msgpack.unpackb(desc, raw=False).get('amdhsa.kernels', [])
    names = [r['.name'] for r in records]
    demangled = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines() \
        if names else []
    out = {}
    for r, d in zip(records, demangled):
        d = d.split('(')[0].replace('void ', '').replace('tsm::', '')
        r = dict(r)
        r.pop('.args', None)
        r['waves_per_simd'] = waves_per_simd(r['.vgpr_count'])
        r['workgroups_per_cu'] = workgroups_per_cu(r, r['.group_segment_fixed_size'])
        out[d] = r
    return out


def waves_per_simd(vgprs: int) -> int:
    alloc = max(VGPR_GRANULE, (vgprs + VGPR_GRANULE - 1) // VGPR_GRANULE * VGPR_GRANULE)
    return min(MAX_WAVES_PER_SIMD, SIMD_VGPRS // alloc)


def workgroups_per_cu(rec: dict, lds_bytes: int) -> int:
    """Resident workgroups of ``.max_flat_workgroup_size`` threads per CU, from registers and LDS (``lds_bytes``: the
    static figure, or static + the dynamic bytes a launch asks for)."""
    waves = (rec['.max_flat_workgroup_size'] + 63) // 64
    by_regs = waves_per_simd(rec['.vgpr_count']) * SIMDS_PER_CU // waves
    by_lds = LDS_PER_CU // lds_bytes if lds_bytes > 0 else 1 << 30
    return max(0, min(by_regs, by_lds))


if __name__ == '__main__':
    md = kernel_metadata()
    print(f'{"vgpr":>5} {"agpr":>5} {"sgpr":>5} {"LDS":>7} {"scratch":>7} {"w/SIMD":>6}  kernel')
    for name, r in sorted(md.items()):
        print(f'{r[".vgpr_count"]:>5} {r[".agpr_count"]:>5} {r[".sgpr_count"]:>5} {r[".group_segment_fixed_size"]:>7} '
              f'{r[".private_segment_fixed_size"]:>7} {r["waves_per_simd"]:>6}  {name}')
