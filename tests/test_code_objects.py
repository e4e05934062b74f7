"""Resource budgets of the built gfx950 code object: registers, LDS, scratch and the occupancy figures
DESIGN.md quotes are read from libtsm_hip.so's own metadata (workoutdetector_amd/codeobj.py) and asserted here, on the CPU.
Several kernels sit right at a budget; a compiler bump that spills inside a K loop or drops a resident workgroup must
fail a test, not show up as an unexplained benchmark regression."""
import pytest

from workoutdetector_amd import codeobj
from workoutdetector_amd.build import build_library


@pytest.fixture(scope='module')
def md():
    return codeobj.kernel_metadata(build_library())


def _one(md, name):
    assert name in md, f'{name} not in the code object; kernels: {sorted(md)[:8]}...'
    return md[name]


def test_every_kernel_is_there_and_wave64(md):
    assert len(md) >= 100
    for name, r in md.items():
        assert r['.wavefront_size'] == 64, name
        assert not r['.uses_dynamic_stack'], name
        assert r['.vgpr_count'] <= codeobj.SIMD_VGPRS, name


def test_no_scratch_anywhere_but_the_two_known_prologue_spills(md):
    """Scratch in a K loop is a 10x slowdown.  The only kernels allowed any are the two segmented fp32 64x64 tiles, which
    trade 1-2 registers spilled OUTSIDE the K loop (8-12 bytes) for the 96-register budget of 5 waves per SIMD (DESIGN
    4.1); everything else -- every MFMA loop of every mode -- must be scratch-free."""
    allowed = {'conv_igemm<64, 64, 2, 2, 3, false, false, 0, false, true>': 16,
               'conv_igemm<64, 64, 2, 2, 1, true, false, 0, false, true>': 16}
    for name, r in md.items():
        limit = allowed.get(name, 0)
        assert r['.private_segment_fixed_size'] <= limit, (name, r['.private_segment_fixed_size'])
        assert r['.vgpr_spill_count'] <= limit // 4, (name, r['.vgpr_spill_count'])     # (SGPRs spill to VGPR lanes, not memory)


def test_fp32_headline_kernel_keeps_five_workgroups_per_cu(md):
    """The dominant kernel of the headline (3x3 convs of layer2-4, segmented K, register-resident K-step): 18 KB of LDS
    and <= 102 registers -> 5 workgroups of 4 waves per CU = 5 waves per SIMD."""
    for name in ('conv_igemm<64, 64, 2, 2, 3, false, false, 0, false, true>',       # 3x3
                 'conv_igemm<64, 64, 2, 2, 1, true, false, 0, false, true>',        # shifted conv1, long K
                 'conv_igemm<64, 64, 2, 2, 1, false, false, 0, true, true>'):       # conv3 + downsample of layer4.0
        r = _one(md, name)
        assert r['.group_segment_fixed_size'] == 18432 and r['.max_flat_workgroup_size'] == 256
        assert r['waves_per_simd'] >= 5 and r['workgroups_per_cu'] >= 5, (name, r['.vgpr_count'])


def test_fused_conv23_kernels_keep_sixteen_waves_per_cu(md):
    for cmid, lds in ((64, 35840), (128, 70656)):
        for x3 in ('false', 'true'):
            r = _one(md, f'conv23_fused_kernel<{cmid}, {x3}>')
            assert r['.group_segment_fixed_size'] == lds and r['.vgpr_count'] <= 128
            waves = r['.max_flat_workgroup_size'] // 64
            assert r['workgroups_per_cu'] * waves >= 16, (cmid, x3, r['workgroups_per_cu'])


def test_bf16_256_tile_fits_one_eight_wave_workgroup(md):
    """conv_bf16_256_kernel: 512 threads = 2 waves per SIMD -> at most 256 registers; 128 KB of dynamic LDS."""
    for args in ('1, false, false, false', '1, true, false, false', '3, false, false, false', '1, false, true, false',
                 '1, false, false, true'):
        r = _one(md, f'conv_bf16_256_kernel<{args}>')
        assert r['.max_flat_workgroup_size'] == 512 and r['.vgpr_count'] <= 256 and r['.group_segment_fixed_size'] == 0
        assert codeobj.workgroups_per_cu(r, 131072) == 1


def test_persistent_bf16_256_tile_budget(md):
    """conv_bf16_256p_kernel: the same 8-wave workgroup (<= 256 registers per wave at 2 waves per SIMD) with the
    accumulators, two tiles' loader state and -- residual arm -- two slabs of prefetched residual live together; NO scratch
    (its counted vmcnt waits assume exactly the vector-memory operations the source issues); 128 KB + the bias of up to
    2048 channels of dynamic LDS = one workgroup per CU."""
    for args in ('1, false, false, false', '1, true, false, false', '3, false, false, false', '1, false, true, false',
                 '1, false, false, true'):
        r = _one(md, f'conv_bf16_256p_kernel<{args}>')
        assert r['.max_flat_workgroup_size'] == 512 and r['.vgpr_count'] <= 256 and r['.group_segment_fixed_size'] == 0
        assert r['.private_segment_fixed_size'] == 0 and r['.vgpr_spill_count'] == 0
        assert codeobj.workgroups_per_cu(r, 131072 + 8192 + 8 * 2176) == 1


def test_weight_stationary_kernels_own_a_whole_register_file(md):
    """One wave per SIMD, up to 512 unified registers each (weights pinned in the accumulation half), no scratch; the
    dynamic LDS they are launched with (csrc constants, static_assert'ed <= 160 KiB at compile time) leaves exactly one
    workgroup per CU."""
    # (accumulation registers: the pinned weights only -- since round 4 these files are built with MFMA results in
    #  architectural VGPRs, build.EXTRA_FLAGS, so the accumulators no longer count here)
    for name, lds, agprs in (('conv3x3_ws_kernel<false>', 2 * 46 * 1024, 200), ('conv3x3_ws_kernel<true>', 162048, 200),
                             ('conv3x3_ws128_kernel<false>', 150000, 200), ('conv3x3_ws128_kernel<true>', 2 * 8 * 289 * 32 + 512, 200),
                             ('conv1x1_wsn_kernel<512, 256, false, 1>', 2 * 65536 + 1024, 160),
                             ('conv1x1_wsn_kernel<384, 256, true, 2>', 2 * 24 * 2048 + 1024, 96)):
        r = _one(md, name)
        assert r['.max_flat_workgroup_size'] == 256 and r['.vgpr_count'] <= 512 and r['.agpr_count'] >= agprs
        assert r['.private_segment_fixed_size'] == 0
        assert codeobj.workgroups_per_cu(r, lds) == 1, name
    assert _one(md, 'conv3x3_ws_kernel<true>')['.vgpr_count'] <= 512


def test_stems_occupancy(md):
    """bf16 stem + max-pool: 77 760 B of LDS and <= 128 registers -> TWO 8-wave workgroups per CU; the
    split-bf16 and fp32 forms hold one (their 32-bit conv tile + weights need > 80 KB)."""
    for name in ('stem_pool_kernel<false, false>', 'stem_pool_kernel<false, true>'):      # (<.., true>: reads [N, 3, H, W] fp32 itself)
        r = _one(md, name)
        assert r['.group_segment_fixed_size'] <= 80 * 1024 and r['.vgpr_count'] <= 128 and r['workgroups_per_cu'] == 2, name
        assert r['.private_segment_fixed_size'] == 0, name
    for name in ('stem_pool_kernel<true, false>', 'stem_pool_kernel<true, true>', 'stem_pool_f32_kernel<false>', 'stem_pool_f32_kernel<true>'):
        r = _one(md, name)
        assert r['.group_segment_fixed_size'] <= codeobj.LDS_PER_CU and r['.vgpr_count'] <= 256
        assert r['workgroups_per_cu'] == 1


def test_whole_bottleneck_kernel_budget(md):
    """bneck_ws_kernel: 272 resident weight registers + the step's working set inside one wave's 512, NO scratch (a spill in
    its step loop would also break its counted vmcnt waits: scratch accesses count as vector-memory operations), and
    150 016 B of dynamic LDS = one workgroup per CU."""
    for cin, lds in ((256, 150016), (64, 133632)):
        for shift, idl in (('true', 'false'), ('false', 'false')) + ((('true', 'true'), ('false', 'true')) if cin == 256 else ()):
            r = _one(md, f'bneck_ws_kernel<{cin}, {shift}, {idl}>')     # (<.., true>: row 2s of the identity from the LDS input slots)
            assert r['.max_flat_workgroup_size'] == 256 and r['.vgpr_count'] <= 512 and r['.agpr_count'] >= 160
            assert r['.vgpr_count'] - r['.agpr_count'] <= 256        # architectural VGPRs incl. the MFMA results (vgpr-form build)
            assert r['.private_segment_fixed_size'] == 0 and r['.vgpr_spill_count'] == 0
            assert codeobj.workgroups_per_cu(r, lds) == 1


def test_front_of_layer2_0_kernel_budget(md):
    """front_s2_kernel (conv1 + stride-2 conv2 of layer2.0 in one launch): 16 + 72 resident weight fragments and the two
    GEMMs' working sets inside one wave's 512 registers, NO scratch (its counted vmcnt waits name exactly the LDS-DMA pieces
    the source issues), 151 552 B of dynamic LDS (3 input-row slots + the 3-row line buffer) = one workgroup per CU."""
    for shift in ('true', 'false'):
        r = _one(md, f'front_s2_kernel<{shift}>')
        assert r['.max_flat_workgroup_size'] == 256 and r['.vgpr_count'] <= 512 and r['.agpr_count'] >= 200
        assert r['.vgpr_count'] - r['.agpr_count'] <= 256
        assert r['.private_segment_fixed_size'] == 0 and r['.vgpr_spill_count'] == 0
        assert codeobj.workgroups_per_cu(r, 151552) == 1


def test_cross_block_kernel_budget_and_wait_tables(md):
    """conv31_fused_kernel: 8 waves = two per SIMD -> at most 256 registers, NO scratch (a spill would be a vector-memory
    operation its counted waits know nothing about); the dynamic LDS of every instantiation fits one workgroup per CU.  And
    the wait tables of csrc/tsm_conv31.hip (C31::wait_w3 / wait_w1 / wait_res: how many younger vector-memory operations may
    stay in flight) restated here and checked against a brute-force simulation of the kernel's issue order."""
    for args, lds in (('128, 512, 128, 1', 151168), ('128, 512, 256, 2', 105000), ('256, 1024, 256, 2', 123000)):
        r = _one(md, f'conv31_fused_kernel<{args}>')
        assert r['.max_flat_workgroup_size'] == 512 and r['.vgpr_count'] <= 256
        assert r['.private_segment_fixed_size'] == 0 and r['.vgpr_spill_count'] == 0
        assert codeobj.workgroups_per_cu(r, lds) == 1

    def simulate(NC, NQ, NW1, NW3, AF, NT1S, PT2, STAGE, RD):
        ops = []
        for t in range(3):
            for c in range(NC):
                ops += [(t, c, 'W1', i) for i in range(NW1)] + [(t, c, 'W3', i) for i in range(NW3)]
                if c == NC - 1:
                    ops += [(t, c, 'AF', i) for i in range(AF)]
                for q in range(NQ):
                    ops += [(t, c, 'st', q), (t, c, 'ld', q)]
                if STAGE and c < NC // 2:
                    ops += [(t, c, 'T2', i) for i in range(PT2)]
                if c == NC - 1:
                    ops += [(t, c, 'T1', i) for i in range(NT1S)]
        at = {o: i for i, o in enumerate(ops)}
        out = {}
        for c in range(NC):
            pt, pc = (1, c - 1) if c > 0 else (0, NC - 1)
            w3 = at[(1, c, 'W1', 0)] - 1 - at[(pt, pc, 'W3', NW3 - 1)]
            end = at[(1, c, 'T2', PT2 - 1)] if (STAGE and c < NC // 2) else at[(1, c, 'ld', NQ - 1)]
            w1 = end - at[(1, c, 'W1', NW1 - 1)]
            res = {at[(1, c, 'st', q)] - 1 - at[divmod(NC + c - RD, NC) + ('ld', q)] for q in range(NQ)}
            assert len(res) == 1
            out[c] = (w3, w1, res.pop())
        return out

    def table(NC, NQ, NW1, NW3, AF, NT1S, PT2, STAGE, RD):
        P = lambda c: PT2 if (STAGE and c < NC // 2) else 0                                     # noqa: E731
        tot = lambda c: NW1 + NW3 + 2 * NQ + P(c) + (AF + NT1S if c == NC - 1 else 0)          # noqa: E731
        out = {}
        for nc in range(NC):
            c0 = (nc - RD + NC) % NC
            n = 2 * (NQ - 1) + P(c0) + (NT1S if c0 == NC - 1 else 0) + NW1 + NW3 + (AF if nc == NC - 1 else 0)
            n += sum(tot((c0 + k) % NC) for k in range(1, RD))
            out[nc] = (AF + 2 * NQ + NT1S if nc == 0 else 2 * NQ + P(nc - 1), NW3 + (AF if nc == NC - 1 else 0) + 2 * NQ + P(nc), n)
        return out

    for cfg in (dict(NC=8, NQ=4, NW1=2, NW3=2, AF=0, NT1S=8, PT2=2, STAGE=True, RD=1),        # <128, 512, 128, 1>
                dict(NC=8, NQ=2, NW1=4, NW3=2, AF=8, NT1S=8, PT2=0, STAGE=False, RD=2),       # <128, 512, 256, 2>
                dict(NC=16, NQ=2, NW1=4, NW3=4, AF=16, NT1S=8, PT2=0, STAGE=False, RD=2)):    # <256, 1024, 256, 2>
        got = table(**cfg)
        assert got == simulate(**cfg), cfg
        assert max(max(v) for v in got.values()) < 64           # vmcnt is a 6-bit field


def test_producer_consumer_cross_block_kernel_budget_and_waits(md):
    """conv31_pc_kernel (round 5: the 128-row tiles with specialised waves): two waves per SIMD -> at most 256 registers, NO
    scratch; 162 432 / 146 048 B of dynamic LDS = one workgroup per CU.  The producers' residual and t2 loads are issued from
    inline asm (hipcc's own wait insertion must not see them: it cut round 4's two-chunk request stream to one), so their
    counted waits are the ONLY thing between a load and its first use: C31P::wait_res restated here against a brute-force
    simulation of the producers' issue order, for both instantiations and residual depths 2 and 4."""
    for args, lds in (('128, 512, 256', 146048), ('256, 1024, 256', 162432)):
        r = _one(md, f'conv31_pc_kernel<{args}>')
        assert r['.max_flat_workgroup_size'] == 512 and r['.vgpr_count'] <= 256
        assert r['.private_segment_fixed_size'] == 0 and r['.vgpr_spill_count'] == 0
        assert lds <= codeobj.LDS_PER_CU and codeobj.workgroups_per_cu(r, lds) == 1

    def simulate(NC, NQ, AF, RD):
        ops = []
        for t in range(4):
            for c in range(NC):
                if c == NC - 1:
                    ops += [(t, c, 'AF', i) for i in range(AF)]          # the next tile's A fragments, behind barrier B
                for q in range(NQ):
                    ops += [(t, c, 'st', q), (t, c, 'ld', q)]            # store y, request the residual of chunk c + RD
        at = {o: i for i, o in enumerate(ops)}
        out = {}
        for c in range(NC):
            t0, c0 = divmod(2 * NC + c - RD, NC)
            res = {at[(2, c, 'st', q)] - 1 - at[(t0, c0, 'ld', q)] for q in range(NQ)}   # younger than the awaited load, at its use
            assert len(res) == 1
            out[c] = res.pop()
        # ... and the A fragments at the head of the next tile: the 2 NQ epilogue operations of the last chunk are younger
        assert at[(2, 0, 'st', 0)] - 1 - at[(1, NC - 1, 'AF', AF - 1)] == 2 * NQ
        return out

    def table(NC, NQ, AF, RD):
        return {nc: 2 * (NQ - 1) + 2 * NQ * (RD - 1) + sum(AF for k in range(1, RD + 1) if (nc - RD + k + NC) % NC == NC - 1)
                for nc in range(NC)}

    for cfg in (dict(NC=8, NQ=4, AF=8, RD=2), dict(NC=16, NQ=4, AF=16, RD=2), dict(NC=16, NQ=4, AF=16, RD=4), dict(NC=8, NQ=4, AF=8, RD=4)):
        got = table(**cfg)
        assert got == simulate(**cfg), cfg
        assert max(got.values()) < 64                          # vmcnt is a 6-bit field
        # the kernel's two cases: chunks whose window (nc - RD, nc] holds a tile's last chunk, and the rest
        wrap = {nc for nc in range(cfg['NC']) if nc == cfg['NC'] - 1 or nc < cfg['RD'] - 1}
        assert {got[nc] for nc in wrap} == {got[0]} and {got[nc] for nc in set(range(cfg['NC'])) - wrap} == {got[cfg['RD'] - 1]}
