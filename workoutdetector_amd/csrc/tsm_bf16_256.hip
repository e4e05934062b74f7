// conv_bf16_256[p]_kernel: the bf16 implicit GEMM on a 256 x 256 tile, operands by LDS-DMA (one-shot and persistent).
#include "tsm_device.h"

namespace tsm {

// ---------------------------------------------------------------------------------------------
// conv_bf16_256: the bf16 implicit GEMM on a 256 x 256 tile, ONE 8-wave workgroup per CU, operands staged by LDS-DMA.
//
// Why: the 128 x 128 bf16 tiles of conv_igemm are LDS-bound -- per MFMA they need as many LDS-array cycles
// (VGPR-staged ds_write_b128 of both operands + fragment reads) as the matrix pipe gives, and one barrier + full
// vmcnt drain per K-step on top (profiles/r01_bf16_pmc_sq_summary.txt: MFMA busy 0.27-0.34).  This kernel is the
// structure cdna_hip_programming.md prescribes for that regime:
//   * 256 x 256 output tile, K-tile 64 channels, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 4 x 2 MFMA tiles of
//     v_mfma_f32_32x32x16_bf16: a quarter of the staging bytes and 3/4 of the fragment reads per MFMA of the 128^2 tile;
//   * both operands go global -> LDS by `buffer_load ... lds` (no VGPR round trip, no ds_write): a wave-instruction
//     writes 1 KiB = 16 rows x 64 B; the 16-byte chunk a lane fetches is XOR-swizzled on the SOURCE side with
//     f(row) = (row >> 2) & 3 and the fragment reads apply the same involution: ds_read_b128 is conflict-free;
//   * a K-tile is cut in four 16-KB "half-operands" by K, not by rows -- {A, B} x {channels 0-31, 32-63} -- because the
//     four phases of a K-tile each multiply ONE k16 group (6 ds_read_b128 + 8 MFMAs per wave): the k 0-31 halves are
//     dead after phase 1 and are re-filled (for K-tile t+2) in phases 2 and 3, the k 32-63 halves in phases 0 and 1 of
//     the next K-tile.  Two 64-KB buffers, FOUR half-operands always in flight, retired by a counted
//     `s_waitcnt vmcnt(8)` twice per K-tile -- never vmcnt(0) inside the loop -- and raw s_barriers;
//   * the two waves of a SIMD run STAGGERED by one barrier (waves 4-7 behind waves 0-3): one is in its 8-MFMA cluster
//     while the other reads fragments and issues DMA, instead of both queueing on the matrix pipe together;
//   * K-tiles past the end of K are staged with an out-of-range offset (zeros, no memory traffic), so the loop and
//     its wait counts are branch-free.
// Per output the products enter the accumulator in conv_igemm's order (k16 groups ascending), so results are
// bit-identical to the other bf16 tiles.  Epilogue: accumulators -> wave-private LDS slab (no workgroup barrier)
// -> + bias, ReLU, bf16, 16-byte stores of whole 128-byte row segments.
// Needs Cout % 256 == 0 and C % 64 == 0.  Template arms: KS = 3; KS = 1 with the fused temporal shift (conv1), with a
// residual (conv3) or with the K-concatenated second source (conv3 + downsample).  The tuner picks it where it wins
// (at least about one tile per CU; long K helps: conv2 and conv1 of layer3-4 at the config-5 size).
// ---------------------------------------------------------------------------------------------
template <int KS, bool SHIFT, bool RES = false, bool DUAL = false>
__global__ void __launch_bounds__(512, 1) conv_bf16_256_kernel(const ConvParams p) {
  static_assert(KS == 1 || KS == 3, "1x1 (optionally temporally shifted) and 3x3");
  static_assert(!(RES || DUAL) || (KS == 1 && !SHIFT), "residual / K-concatenated second source: plain 1x1 convs (conv3)");
  static_assert(!(RES && DUAL), "the fused conv3 + downsample GEMM has no residual");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 buffers x 64 KB; epilogue: 8 x 8704 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5, l31 = lane & 31;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (p.reverse) tile = nwg - 1 - tile;
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * 256, n0 = tn * 256;

  const int HoWo = p.Ho * p.Wo;
  const int n_first = m0 / HoWo;
  const int frame0 = SHIFT ? (n_first > 0 ? n_first - 1 : 0) : n_first;
  const int frame_bytes = p.Hi * p.Wi * p.C * 2;
  const size_t a_bytes = ((size_t)p.N - frame0) * (size_t)frame_bytes;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)frame0 * frame_bytes), 0,
      (int)(a_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(p.w) + (size_t)n0 * p.Kp * 2), 0, 256 * p.Kp * 2, 0x00020000);

  // second A source (DUAL: conv3 + downsample as one GEMM, K = [conv3 input channels | block input channels])
  const int frame_bytes2 = DUAL ? p.Hi2 * p.Wi2 * p.C2 * 2 : 0;
  const size_t a2_bytes = DUAL ? ((size_t)p.N - n_first) * (size_t)frame_bytes2 : 0;
  const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(DUAL ? p.x2 : p.x) + (size_t)n_first * frame_bytes2), 0,
      (int)(a2_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a2_bytes), 0x00020000);
  const int nt1 = DUAL ? p.K1 / 64 : 0;                   // K-tiles of the first source

  // ---- loader state: this lane fills LDS slot (row, lane & 3) of rows piece * 16 + (lane >> 2), piece = 2 * wave + q
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);      // global 16-B chunk held by that slot (swizzle on the source)
  unsigned a_off[2], a_offp[SHIFT ? 2 : 1], a_offm[SHIFT ? 2 : 1], a_mask[KS == 3 ? 2 : 1], b_off[2], a_off2[DUAL ? 2 : 1];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (2 * wave + q) * 16 + (lane >> 2);
    const int m = m0 + row;
    const bool ok = m < p.M;
    const int mm = ok ? m : m0;
    const int n = mm / HoWo, rem = mm - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const int base = (n - frame0) * frame_bytes + (iy0 * p.Wi + ix0) * p.C * 2 + chunk * 16;
    a_off[q] = (KS == 1 && !ok) ? kInvalid : (unsigned)base;
    if (KS == 3) {
      unsigned mask = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)p.Hi && (unsigned)(ix0 + kx) < (unsigned)p.Wi) mask |= 1u << (ky * 3 + kx);
      a_mask[q] = ok ? mask : 0u;
    }
    if (SHIFT) {
      const int t = n % p.T;
      a_offp[q] = (ok && t < p.T - 1) ? (unsigned)(base + frame_bytes) : kInvalid;
      a_offm[q] = (ok && t > 0) ? (unsigned)(base - frame_bytes) : kInvalid;
    }
    if (DUAL)
      a_off2[q] = ok ? (unsigned)((n - n_first) * frame_bytes2 + (oy * p.stride2 * p.Wi2 + ox * p.stride2) * p.C2 * 2 + chunk * 16)
                     : kInvalid;
    b_off[q] = (unsigned)(row * p.Kp * 2 + chunk * 16);
  }
  const int nt = p.Kp / 64;                               // K-tiles
  typedef __attribute__((address_space(3))) void lds_void;
  // Stage one half-operand of K-tile kt: which = 0 A k0-31, 1 B k0-31, 2 A k32-63, 3 B k32-63 (two 1-KiB pieces per wave)
  auto stage = [&](int kt, int which) {
    const unsigned dead = (~(unsigned)((kt - nt) >> 31)) & kInvalid;
    const int kh = which >> 1;
    const unsigned kbytes = (unsigned)kt * 128u + (unsigned)kh * 64u;
    unsigned char *dst = lds + (kt & 1) * 65536 + ((which & 1) * 2 + kh) * 16384 + wave * 2048;
    if (which & 1) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void *)(dst + q * 1024), 16, (int)(b_off[q] | dead), (int)kbytes, 0, 0);
    } else if (KS == 1) {
      unsigned mp = 0u, mm_ = 0u, m0_ = ~0u;
      if (SHIFT) {
        const int c = kt * 64 + kh * 32 + chunk * 8;      // first channel of this lane's chunk
        mp = 0u - (unsigned)(c < p.fold);
        mm_ = (0u - (unsigned)(c < 2 * p.fold)) & ~mp;
        m0_ = ~(mp | mm_);
      }
      const bool second = DUAL && kt >= nt1;               // wave-uniform: which source this K-tile comes from
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned off = a_off[q];
        if (SHIFT) off = (a_offp[q] & mp) | (a_offm[q] & mm_) | (a_off[q] & m0_);
        if (DUAL && second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void *)(dst + q * 1024), 16, (int)(a_off2[q] | dead),
                                                   (int)(kbytes - (unsigned)nt1 * 128u), 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void *)(dst + q * 1024), 16, (int)(off | dead), (int)kbytes, 0, 0);
      }
    } else {
      const int tap = (kt * 64) >> (p.logC4 + 2);         // C >= 64: a K-tile never straddles a tap
      const int ky = tap / 3, kx = tap - ky * 3;
      const unsigned tap_off = (unsigned)(((ky * p.Wi + kx) * p.C + (kt * 64 - tap * p.C) + kh * 32) * 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void *)(dst + q * 1024), 16,
                                                 (int)((((a_mask[q] >> tap) & 1u) ? a_off[q] + tap_off : kInvalid) | dead), 0, 0, 0);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: row * 64 B + swizzled chunk; k16 group g reads chunk 2 * (g & 1) + half of region g >> 1
  const int sw = (l31 >> 2) & 3;
  const unsigned a_rd0 = (unsigned)((wm * 128 + l31) * 64 + ((0 + half) ^ sw) * 16);
  const unsigned a_rd1 = (unsigned)((wm * 128 + l31) * 64 + ((2 + half) ^ sw) * 16);
  const unsigned b_rd0 = (unsigned)(32768 + (wn * 64 + l31) * 64 + ((0 + half) ^ sw) * 16);
  const unsigned b_rd1 = (unsigned)(32768 + (wn * 64 + l31) * 64 + ((2 + half) ^ sw) * 16);

  // prologue: the six half-operands the schedule has in flight before K-tile 0 starts
  stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3); stage(1, 0); stage(1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // K-tile 0, k 0-31 of A and B have landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  // Stagger: waves 4-7 (wm == 1; the second wave of every SIMD) run one barrier behind waves 0-3, so that on each
  // SIMD one wave is in its MFMA cluster while its partner reads fragments / issues DMA -- in lockstep both would
  // read together and then queue on the one matrix pipe (MI355X_MICROARCH.md, two waves per SIMD, item 9).
  // Consequences for the hand-placed synchronisation: (a) the counted vmcnt sits BEFORE the first barrier of the odd
  // phases, so that the delayed group too has retired its DMA one barrier before the early group reads the data;
  // (b) the fragment reads are retired (lgkmcnt(0)) before the first barrier of their phase, so that the DMA which the
  // early group issues one phase later cannot overtake a read of the delayed group.
  if (wm == 1) __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nt; ++kt) {
    const unsigned buf = (unsigned)(kt & 1) * 65536u;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      u32x4 af[4], bf[2];
      {
        const unsigned ra = buf + (ph >> 1) * 16384u + ((ph & 1) ? a_rd1 : a_rd0);
        const unsigned rb = buf + (ph >> 1) * 16384u + ((ph & 1) ? b_rd1 : b_rd0);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4 *>(lds + rb + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4 *>(lds + ra + i * 2048);
      }
      // refill what the previous phases have finished reading: k 32-63 halves of the OTHER buffer (K-tile kt+1) in
      // phases 0-1, k 0-31 halves of THIS buffer (K-tile kt+2) in phases 2-3
      if (ph == 0) stage(kt + 1, 2);
      else if (ph == 1) stage(kt + 1, 3);
      else if (ph == 2) stage(kt + 2, 0);
      else stage(kt + 2, 1);
      // odd phases: the counted wait that retires the two half-operands the NEXT phase reads (8 = the four younger
      // half-operands x 2 pieces per wave stay in flight; never 0 inside the loop)
      if (ph & 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
                                                              acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();              // the early group waits for the delayed one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dead tail stages (zeros) must land before LDS is reused
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: per wave, four 32 x 64 slabs through a private LDS region ---------------------------------
  float *Cs = reinterpret_cast<float *>(lds + wave * 8704);  // [32][68] fp32
  const size_t y_bytes = ((size_t)p.M - m0) * p.Cout * 2;
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.y) + (size_t)m0 * p.Cout * 2, 0, (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes), 0x00020000);
  const float floor_ = p.relu ? 0.f : -INFINITY;
  const int c8 = lane & 7, r8l = lane >> 3;
  const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 64 + c8 * 8);
  const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 64 + c8 * 8 + 4);
  // residual (RES): the 8 channels of this lane's row segment, fetched one slab ahead of their use
  const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(RES ? p.res : p.y) + (size_t)m0 * p.Cout * 2), 0,
      (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes), 0x00020000);
  u32x4 rres[2][RES ? 4 : 1];
  auto load_res = [&](int i, int set) {
    if constexpr (RES) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        rres[set][k] = __builtin_amdgcn_raw_buffer_load_b128(
            rsrcR, ((wm * 128 + i * 32 + r8l + 8 * k) * p.Cout + n0 + wn * 64 + c8 * 8) * 2, 0, 0);
    }
  };
  load_res(0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i + 1 < 4) load_res(i + 1, (i + 1) & 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) Cs[((e & 3) + 8 * (e >> 2) + 4 * half) * 68 + j * 32 + l31] = acc[i][j][e];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (same wave wrote it: no barrier needed)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = r8l + 8 * k;
      const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cs + rr * 68 + c8 * 8);
      const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cs + rr * 68 + c8 * 8 + 4);
      float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                    c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
      if constexpr (RES) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += split_elem(rres[i & 1][k], e);
      }
      u32x4 o;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) o[w2] = pack_bf16(fmaxf(v[2 * w2], floor_), fmaxf(v[2 * w2 + 1], floor_));
      const int row = wm * 128 + i * 32 + rr;
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (row * p.Cout + n0 + wn * 64 + c8 * 8) * 2, 0, TSM_AUX_256);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slab reads done before the next slab overwrites it
  }
}

bool conv_bf16_256_valid(const ConvParams &p, int ks) {
  if (p.prec != kPrecBf16 || (ks != 1 && ks != 3) || p.Cout % 256 != 0 || p.C % 64 != 0 || p.Kp % 64 != 0 || p.kseg_len != 0)
    return false;
  if (ks == 3) return !p.res && !p.x2 && p.T == 0;
  if (p.T > 0) return !p.res && !p.x2;                    // shifted conv1
  if (p.x2) return !p.res && p.K1 % 64 == 0 && p.C2 % 64 == 0;
  return true;
}

// ---------------------------------------------------------------------------------------------
// conv_bf16_256p: conv_bf16_256's K pipeline run PERSISTENTLY over the tiles of a workgroup, without ever draining.
//
// Why: conv_bf16_256 is one 128-KB workgroup per CU, so nothing overlaps a tile's prologue (the first operands' HBM
// latency, 2-3 us) or its epilogue (accumulators through LDS, residual loads, stores) -- and with short K that is a
// third of a tile (K = 256: four K-tiles = 3.4 us of MFMA per 17-us tile; the 1x1 launches of layer3/4 sit at 3.0-4.3
// TB/s and 0.29-0.44 of the MFMA peak: neither roofline).  Here
//   * the K-tile sequence is FLAT across tiles: the refills the last two K-tiles of tile s issue ("K-tile kt + 1,
//     kt + 2") are the first K-tiles of tile s + 1, staged from that tile's loader state (two sets of lane offsets and
//     descriptors, current / next); the four-half-operands-in-flight schedule, its counted waits, the staggered wave
//     groups and the barriers are conv_bf16_256's, unchanged, and run from the first K-tile of the first tile to the
//     last K-tile of the last;
//   * the epilogue leaves the two operand buffers alone, so it can sit between two K-tiles while the next tile's operands
//     land.  Without a residual it touches no LDS at all: the product is TRANSPOSED (A = weights, B = pixels: the fragment
//     formats are symmetric, the products of an output enter its accumulator in the same k order -> same bits), a lane
//     then holds 4-channel runs of ONE pixel, and bias + ReLU + bf16 + v_permlane32_swap give 16-byte stores straight
//     from registers (conv3x3_ws's epilogue; measured against the slab form below: 3-6 % faster on these arms).  With a
//     residual the product is not transposed and the epilogue goes through eight wave-private [8][68] fp32 SUB-SLABS
//     behind the bias (whole 128-byte row segments of residual and output per 8 lanes; the register form, 16 bytes per
//     lane and two lanes per pixel, lost to conv_bf16_256 on exactly these launches), residual eight sub-slabs ahead;
//     the bias of all Cout channels sits in LDS behind the two buffers;
//   * the 16 stores of an epilogue are younger than the operands the next K-tile waits for: its two counted waits
//     are vmcnt(8 + 16) instead of vmcnt(8) (vector-memory operations retire in order; a vmcnt(8) there would wait
//     for the stores' completion); the residual arm requests sub-slabs 0-7 at the START of the tile's last K-tile,
//     whose waits are therefore vmcnt(8 + 8).
// Needs at least two K-tiles per tile (K >= 128: "kt + 2" must not skip a tile) and Cout <= 2048 (the bias in LDS).
// Bit-identical to conv_bf16_256 and to conv_igemm's bf16 tiles; the tuner picks per layer.
// ---------------------------------------------------------------------------------------------
#ifndef TSM_256P_DMA_IN_MFMA
#define TSM_256P_DMA_IN_MFMA 0   // measured at config 5: 3x3 launches 250 -> 266 us, conv3 + downsample of layer2.0 540 -> 595 us (anything
#endif                           // added to the MFMA cluster lengthens the critical path: the other group's memory phase is the shorter one)
#ifndef TSM_256P_X
#define TSM_256P_X 0             // timing probes (garbage results): 1 no DMA in the K loop, 2 the DMA reads nothing (dead offsets), 4 one fragment read per step
#endif
#ifndef TSM_256P_PAIR
#define TSM_256P_PAIR 1          // 3x3 arm: two 8-MFMA clusters (k 0-15, k 16-31 of a K-half) per barrier pair instead of one.  Measured at config 5:
                                 // stride-1 3x3 launches unchanged (253 vs 252 us), the stride-2 one of layer3.0 312 -> 288 us; the 1x1 arms lose
                                 // 0-3 % (conv3 + downsample of layer2.0 545 -> 562 us) and keep one cluster per pair
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
#ifndef TSM_256P_STAMP
#define TSM_256P_STAMP 0   // diagnostic builds only: per-phase cycle sums of workgroup 0 (s_memtime), printed at the kernel's end (1: conv3 + downsample, 2: 3x3)
#endif
#if TSM_256P_STAMP
#define P256_STAMP(i)                                       \
  do {                                                      \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += _t - stamp_last;                        \
    stamp_last = _t;                                        \
  } while (0)
#else
#define P256_STAMP(i) do {} while (0)
#endif
constexpr size_t kLds256pBytes = 131072 + 8192 + 8 * 2176;   // two operand buffers | bias [<= 2048] fp32 | residual arm: eight [8][68] fp32 sub-slabs

template <int KS, bool SHIFT, bool DUAL> struct Tile256State {
  unsigned a_off[2], b_off[2];
  unsigned a_mask[KS == 3 ? 2 : 1];
  unsigned a_offp[SHIFT ? 2 : 1], a_offm[SHIFT ? 2 : 1];
  unsigned a_off2[DUAL ? 2 : 1];
  int m0, n0;
  // (the operand windows as plain pointers + sizes: the host pass cannot hold __amdgpu_buffer_rsrc_t in a struct; the
  //  descriptors are rebuilt where they are used -- scalar moves)
  const char *pa, *pb, *pa2;
  int sza, sza2;
};

template <int KS, bool SHIFT, bool RES = false, bool DUAL = false>
__global__ void __launch_bounds__(512, 1) conv_bf16_256p_kernel(const ConvParams p) {
  static_assert(KS == 1 || KS == 3, "1x1 (optionally temporally shifted) and 3x3");
  static_assert(!(RES || DUAL) || (KS == 1 && !SHIFT), "residual / K-concatenated second source: plain 1x1 convs (conv3)");
  static_assert(!(RES && DUAL), "the fused conv3 + downsample GEMM has no residual");
#if TSM_256P_STAMP
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 buffers x 64 KB | bias [Cout] fp32
  typedef __attribute__((address_space(3))) void lds_void;
  typedef Tile256State<KS, SHIFT, DUAL> State;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int half = lane >> 5, l31 = lane & 31;

  const int ntiles = p.ntm * p.ntn, nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int my = (ntiles - bid + nwg - 1) / nwg;          // tiles of this workgroup (>= 1: the grid never exceeds the tiles)
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int HoWo = p.Ho * p.Wo;
  const int frame_bytes = p.Hi * p.Wi * p.C * 2;
  const int frame_bytes2 = DUAL ? p.Hi2 * p.Wi2 * p.C2 * 2 : 0;
  const int nt1 = DUAL ? p.K1 / 64 : 0;                   // K-tiles of the first source
  const int nt = p.Kp / 64;                               // K-tiles per tile (>= 2)
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);       // global 16-B chunk held by this lane's LDS slot (source-side swizzle)

  float *bias_lds = reinterpret_cast<float *>(lds + 131072);
  for (int i = tid; i < p.Cout; i += 512) bias_lds[i] = p.bias[i];

  // the loader state of the s-th tile of this workgroup (virtual block bid + s * nwg in conv_bf16_256's XCD-chunked order)
  auto setup = [&](State &T, int s) {
    const int v = bid + s * nwg, xcd = v & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
    if (tile >= ntiles) tile = ntiles - 1;                // (s == my: never staged live, kept in range for the arithmetic)
    if (p.reverse) tile = ntiles - 1 - tile;
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    T.m0 = tm * 256;
    T.n0 = tn * 256;
    const int n_first = T.m0 / HoWo;
    const int frame0 = SHIFT ? (n_first > 0 ? n_first - 1 : 0) : n_first;
    const size_t a_bytes = ((size_t)p.N - frame0) * (size_t)frame_bytes;
    T.pa = reinterpret_cast<const char *>(p.x) + (size_t)frame0 * frame_bytes;
    T.sza = (int)(a_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a_bytes);
    T.pb = reinterpret_cast<const char *>(p.w) + (size_t)T.n0 * p.Kp * 2;
    const size_t a2_bytes = DUAL ? ((size_t)p.N - n_first) * (size_t)frame_bytes2 : 0;
    T.pa2 = reinterpret_cast<const char *>(DUAL ? p.x2 : p.x) + (size_t)n_first * frame_bytes2;
    T.sza2 = (int)(a2_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a2_bytes);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (2 * wave + q) * 16 + (lane >> 2);
      const int m = T.m0 + row;
      const bool ok = m < p.M;
      const int mm = ok ? m : T.m0;
      const int n = mm / HoWo, rem = mm - n * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      const int base = (n - frame0) * frame_bytes + (iy0 * p.Wi + ix0) * p.C * 2 + chunk * 16;
      T.a_off[q] = (KS == 1 && !ok) ? kInvalid : (unsigned)base;
      if (KS == 3) {
        unsigned mask = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if ((unsigned)(iy0 + ky) < (unsigned)p.Hi && (unsigned)(ix0 + kx) < (unsigned)p.Wi) mask |= 1u << (ky * 3 + kx);
        T.a_mask[q] = ok ? mask : 0u;
      }
      if (SHIFT) {
        const int t = n % p.T;
        T.a_offp[q] = (ok && t < p.T - 1) ? (unsigned)(base + frame_bytes) : kInvalid;
        T.a_offm[q] = (ok && t > 0) ? (unsigned)(base - frame_bytes) : kInvalid;
      }
      if (DUAL)
        T.a_off2[q] = ok ? (unsigned)((n - n_first) * frame_bytes2 + (oy * p.stride2 * p.Wi2 + ox * p.stride2) * p.C2 * 2 + chunk * 16)
                         : kInvalid;
      T.b_off[q] = (unsigned)(row * p.Kp * 2 + chunk * 16);
    }
  };

  // Stage one half-operand of K-tile kt of the tile with state T into buffer `par`: which = 0 A k0-31, 1 B k0-31,
  // 2 A k32-63, 3 B k32-63 (two 1-KiB pieces per wave); dead = kInvalid: zeros, no memory traffic
  auto stage_of = [&](const State &T, int kt, unsigned par, int which, unsigned dead, int qsel = -1) {   // qsel: one of the two pieces, or both
    const int kh = which >> 1;
    const unsigned kbytes = (unsigned)kt * 128u + (unsigned)kh * 64u;
    unsigned char *dst = lds + par * 65536u + ((which & 1) * 2 + kh) * 16384 + wave * 2048;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(T.pa), 0, T.sza, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(T.pb), 0, 256 * p.Kp * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(T.pa2), 0, T.sza2, 0x00020000);
    if (which & 1) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (qsel < 0 || q == qsel)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void *)(dst + q * 1024), 16, (int)(T.b_off[q] | dead), (int)kbytes, 0, 0);
    } else if (KS == 1) {
      unsigned mp = 0u, mm_ = 0u, m0_ = ~0u;
      if (SHIFT) {
        const int c = kt * 64 + kh * 32 + chunk * 8;      // first channel of this lane's chunk
        mp = 0u - (unsigned)(c < p.fold);
        mm_ = (0u - (unsigned)(c < 2 * p.fold)) & ~mp;
        m0_ = ~(mp | mm_);
      }
      const bool second = DUAL && kt >= nt1;               // wave-uniform: which source this K-tile comes from
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (qsel >= 0 && q != qsel) continue;
        unsigned off = T.a_off[q];
        if (SHIFT) off = (T.a_offp[q] & mp) | (T.a_offm[q] & mm_) | (T.a_off[q] & m0_);
        if (DUAL && second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, (lds_void *)(dst + q * 1024), 16, (int)(T.a_off2[q] | dead),
                                                   (int)(kbytes - (unsigned)nt1 * 128u), 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void *)(dst + q * 1024), 16, (int)(off | dead), (int)kbytes, 0, 0);
      }
    } else {
      const int tap = (kt * 64) >> (p.logC4 + 2);         // C >= 64: a K-tile never straddles a tap
      const int ky = tap / 3, kx = tap - ky * 3;
      const unsigned tap_off = (unsigned)(((ky * p.Wi + kx) * p.C + (kt * 64 - tap * p.C) + kh * 32) * 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (qsel < 0 || q == qsel)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void *)(dst + q * 1024), 16,
                                                 (int)((((T.a_mask[q] >> tap) & 1u) ? T.a_off[q] + tap_off : kInvalid) | dead), 0, 0, 0);
    }
  };

  f32x16 acc[4][2];      // acc[i][j]: rows = channels n0 + 64 wn + 32 j + .., columns = pixels m0 + 128 wm + 32 i + l31
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int sw = (l31 >> 2) & 3;
  const unsigned a_rd0 = (unsigned)((wm * 128 + l31) * 64 + ((0 + half) ^ sw) * 16);
  const unsigned a_rd1 = (unsigned)((wm * 128 + l31) * 64 + ((2 + half) ^ sw) * 16);
  const unsigned b_rd0 = (unsigned)(32768 + (wn * 64 + l31) * 64 + ((0 + half) ^ sw) * 16);
  const unsigned b_rd1 = (unsigned)(32768 + (wn * 64 + l31) * 64 + ((2 + half) ^ sw) * 16);
  const float floor_ = p.relu ? 0.f : -INFINITY;

  // Residual arm (RES): the product is NOT transposed and the epilogue goes through LDS like conv_bf16_256's -- whole
  // 128-byte row segments of the residual and of the output per 8 lanes; the transposed register epilogue would fetch the
  // residual as 16 bytes per lane, two lanes per pixel, and lost to conv_bf16_256 on exactly these launches -- but in
  // SUB-SLABS of 8 rows x 64 channels ([8][68] fp32 per wave = 17 KB for the workgroup, behind the bias): the two operand
  // buffers stay untouched, so the next tile's operands still land under the epilogue.  A wave tile is 16 sub-slabs; the
  // residual of sub-slab t is ONE 16-byte load per lane (row t * 8 + lane / 8, channels 8 (lane % 8) ..), eight of
  // them in flight: sub-slabs 0-7 are requested at the START of the tile's last K-tile -- older than that K-tile's four
  // operand stages, so that consuming them does not wait for the next tile's operands (vector-memory operations retire
  // in order) -- and sub-slab t + 8 when sub-slab t has been consumed.
  u32x4 rres[RES ? 8 : 1];
  const int c8 = lane & 7, r8l = lane >> 3;
  auto load_res = [&](const State &T, int t, int slot) {
    if constexpr (RES) {
      const size_t y_bytes = ((size_t)p.M - T.m0) * p.Cout * 2;
      const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.res) + (size_t)T.m0 * p.Cout * 2), 0,
          (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes), 0x00020000);
      rres[slot] = __builtin_amdgcn_raw_buffer_load_b128(
          rsrcR, ((wm * 128 + t * 8 + r8l) * p.Cout + T.n0 + wn * 64 + c8 * 8) * 2, 0, 0);
    }
  };

  State cur, nxt;
  setup(cur, 0);
  // prologue: the six half-operands the schedule has in flight before the first K-tile starts (nt >= 2: all of tile 0)
  stage_of(cur, 0, 0u, 0, 0u); stage_of(cur, 0, 0u, 1, 0u); stage_of(cur, 0, 0u, 2, 0u); stage_of(cur, 0, 0u, 3, 0u);
  stage_of(cur, 1, 1u, 0, 0u); stage_of(cur, 1, 1u, 1, 0u);
  asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // K-tile 0, k 0-31 of A and B have landed (this wave's share); the bias is written
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();              // the stagger of conv_bf16_256: waves 4-7 one barrier behind

  int g = 0;                                              // K-tiles done so far: the LDS buffer of a K-tile is its parity
  for (int s = 0; s < my; ++s) {
    const unsigned next_dead = s + 1 < my ? 0u : kInvalid;
    P256_STAMP(5);
    setup(nxt, s + 1);
    P256_STAMP(0);
    for (int kt = 0; kt < nt; ++kt, ++g) {
      const unsigned buf = (unsigned)(g & 1) * 65536u;
      // K-tile kt + d of the flat sequence: this tile's, or the first ones of the next tile
      auto stage = [&](int d, int which, int qsel) {
        const unsigned par = (unsigned)((g + d) & 1);
        if (kt + d < nt) stage_of(cur, kt + d, par, which, (TSM_256P_X & 2) ? kInvalid : 0u, qsel);
        else stage_of(nxt, kt + d - nt, par, which, next_dead, qsel);
      };
      auto stage_ph = [&](int ph, int qsel) {
        if (ph == 0) stage(1, 2, qsel);
        else if (ph == 1) stage(1, 3, qsel);
        else if (ph == 2) stage(2, 0, qsel);
        else stage(2, 1, qsel);
      };
      const bool after_epilogue = kt == 0 && s > 0;       // 16 stores sit between the operands awaited here and the younger DMA
      const bool with_res = RES && kt == nt - 1;          // 8 residual loads sit there (issued right here)
      if (with_res) {
#pragma unroll
        for (int t = 0; t < 8; ++t) load_res(cur, t, t);
      }
      constexpr bool kPair = TSM_256P_PAIR != 0 && KS == 3;
      if constexpr (kPair) {
        // One barrier pair per K-HALF: the twelve fragment reads and the four DMA pieces of both of its k-16 steps in front of the
        // first barrier, sixteen MFMAs (the same order as below: k 0-15 then k 16-31 of every accumulator) behind it.  The waits are
        // those of the odd phases below, behind the same operations.
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          u32x4 af[2][4], bf[2][2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
#if TSM_256P_X & 4     // timing probe: one fragment read per step instead of six (garbage results)
            if (h2 == 1) { af[1][0] = af[0][0]; af[1][1] = af[0][0]; af[1][2] = af[0][0]; af[1][3] = af[0][0]; bf[1][0] = af[0][0]; bf[1][1] = af[0][0]; continue; }
            af[0][0] = *reinterpret_cast<const u32x4 *>(lds + buf + pp * 16384u + a_rd0);
            af[0][1] = af[0][0]; af[0][2] = af[0][0]; af[0][3] = af[0][0]; bf[0][0] = af[0][0]; bf[0][1] = af[0][0];
            continue;
#endif
            const unsigned ra = buf + pp * 16384u + (h2 ? a_rd1 : a_rd0);
            const unsigned rb = buf + pp * 16384u + (h2 ? b_rd1 : b_rd0);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[h2][j] = *reinterpret_cast<const u32x4 *>(lds + rb + j * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[h2][i] = *reinterpret_cast<const u32x4 *>(lds + ra + i * 2048);
          }
#if TSM_256P_X & 1     // timing probe: no DMA issued in the K loop (garbage results)
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
          stage_ph(2 * pp, -1);
          stage_ph(2 * pp + 1, -1);
          if (after_epilogue) wait_vmcnt_lgkm0<24>();
          else wait_vmcnt_lgkm0<8>();
#endif
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[h2][j]), __builtin_bit_cast(bf16x8, af[h2][i]),
                                                                    acc[i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
      } else {
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        u32x4 af[4], bf[2];
        {
          const unsigned ra = buf + (ph >> 1) * 16384u + ((ph & 1) ? a_rd1 : a_rd0);
          const unsigned rb = buf + (ph >> 1) * 16384u + ((ph & 1) ? b_rd1 : b_rd0);
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4 *>(lds + rb + j * 2048);
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const u32x4 *>(lds + ra + i * 2048);
        }
        // The phase's two DMA pieces: in front of the barrier (the other wave group's MFMAs run meanwhile), or -- TSM_256P_DMA_IN_MFMA --
        // one by one BEHIND this wave's own MFMAs, where issuing a piece costs next to nothing (in a burst in front of a barrier
        // ~100-150 cycles each: the part of a phase that is not MFMA was longer than the other group's MFMAs).  The phase's wait then
        // sees two operations fewer behind the operands it waits for.
        constexpr bool kIn = TSM_256P_DMA_IN_MFMA != 0 && !RES;   // (the residual arm sits at 256 registers: it would spill)
        if (!kIn) stage_ph(ph, -1);
        if (ph & 1) {
          if (after_epilogue) wait_vmcnt_lgkm0<kIn ? 22 : 24>();
          else if (with_res) wait_vmcnt_lgkm0<kIn ? 14 : 16>();
          else wait_vmcnt_lgkm0<kIn ? 6 : 8>();
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = RES ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
                                                                      acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af[i]),
                                                                      acc[i][j], 0, 0, 0);
            if (kIn && j == 1 && (i == 0 || i == 2)) {
              __builtin_amdgcn_sched_barrier(0);
              stage_ph(ph, i >> 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      }
      P256_STAMP((kt == 0 ? 1 : kt == 1 ? 2 : 3));
    }
    // ---- epilogue of tile s: the next tile's first operands are in flight ----
    if constexpr (!RES) {   // from registers (no LDS, no barrier)
      const size_t y_bytes = ((size_t)p.M - cur.m0) * p.Cout * 2;
      const int ysz = (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes);
      const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char *>(p.y) + (size_t)cur.m0 * p.Cout * 2, 0, ysz, 0x00020000);
      const int cbase = (cur.n0 + wn * 64) * 2 + half * 32;               // this lane's first byte within a pixel's row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = (wm * 128 + i * 32 + l31) * p.Cout * 2 + cbase;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned pk[4][2];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + cur.n0 + wn * 64 + j * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
              float v0 = acc[i][j][4 * q + 2 * w2] + b[2 * w2], v1 = acc[i][j][4 * q + 2 * w2 + 1] + b[2 * w2 + 1];
              pk[q][w2] = pack_bf16(fmaxf(v0, floor_), fmaxf(v1, floor_));
            }
          }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
              const auto r2 = __builtin_amdgcn_permlane32_swap(pk[qq][w2], pk[qq + 2][w2], false, false);
              pk[qq][w2] = r2[0];
              pk[qq + 2][w2] = r2[1];
            }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const u32x4 ov = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
#ifdef TSM_256P_NOSTORE   // probe: the stores fall outside the window (no memory traffic)
            __builtin_amdgcn_raw_buffer_store_b128(ov, rsrcY, (int)((unsigned)o | kInvalid), j * 64 + qq * 16, TSM_AUX_256);
#else
            __builtin_amdgcn_raw_buffer_store_b128(ov, rsrcY, o, j * 64 + qq * 16, TSM_AUX_256);
#endif
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
      }
    } else {                // residual arm: 16 sub-slabs of 8 rows through this wave's [8][68] fp32 slab
      float *Cs = reinterpret_cast<float *>(lds + 131072 + 8192 + wave * 2176);
      const size_t y_bytes = ((size_t)p.M - cur.m0) * p.Cout * 2;
      const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char *>(p.y) + (size_t)cur.m0 * p.Cout * 2, 0, (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes), 0x00020000);
      const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(bias_lds + cur.n0 + wn * 64 + c8 * 8);
      const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(bias_lds + cur.n0 + wn * 64 + c8 * 8 + 4);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = t >> 2, q = t & 3;          // rows 32 i + 8 q .. + 8 of the wave tile: accumulator elements 4 q .. 4 q + 3
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cs[(4 * half + r) * 68 + j * 32 + l31] = acc[i][j][4 * q + r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (same wave wrote it: no barrier needed)
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cs + r8l * 68 + c8 * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cs + r8l * 68 + c8 * 8 + 4);
        float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                      c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += split_elem(rres[t & 7], e);
        u32x4 o;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) o[w2] = pack_bf16(fmaxf(v[2 * w2], floor_), fmaxf(v[2 * w2 + 1], floor_));
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, ((wm * 128 + t * 8 + r8l) * p.Cout + cur.n0 + wn * 64 + c8 * 8) * 2, 0, TSM_AUX_256);
        if (t + 8 < 16) load_res(cur, t + 8, t & 7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // slab reads done before the next sub-slab overwrites it
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    P256_STAMP(4);
    cur = nxt;
  }
#if TSM_256P_STAMP
  if (((TSM_256P_STAMP == 1 && DUAL) || (TSM_256P_STAMP == 2 && KS == 3)) && bid == 0 && (tid == 0 || tid == 256))
    printf("256p KS=%d DUAL=%d wave %d tiles=%d nt=%d: setup %llu ktile0 %llu ktile1 %llu ktiles2+ %llu epilogue %llu other %llu cycles\n", KS, (int)DUAL,
           wave, my, nt, stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3], stamp_acc[4], stamp_acc[5]);
#endif
  if (wm == 0) __builtin_amdgcn_s_barrier();              // the early group waits for the delayed one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dead tail stages (zeros) land before the workgroup leaves its LDS
}

bool conv_bf16_256p_valid(const ConvParams &p, int ks) {
  return conv_bf16_256_valid(p, ks) && p.Kp >= 128 && p.Cout <= 2048;
}

constexpr size_t kLds256Bytes = 131072;

hipError_t launch_conv_bf16_256(ConvParams p, int ks, hipStream_t s) {
  if (!conv_bf16_256_valid(p, ks)) return hipErrorInvalidValue;
  p.ntm = (p.M + 255) / 256;
  p.ntn = p.Cout / 256;
  const dim3 grid((unsigned)(p.ntm * p.ntn)), block(512);
  constexpr size_t kLdsBytes = kLds256Bytes;
  const DeviceInfo &di = device_info();   // the > 64 KB dynamic-LDS opt-in, once per device
  if (di.status != hipSuccess) return di.status;
  if (ks == 3) TSM_KLAUNCH((conv_bf16_256_kernel<3, false>), grid, block, kLdsBytes, s, p);
  else if (p.T > 0) TSM_KLAUNCH((conv_bf16_256_kernel<1, true>), grid, block, kLdsBytes, s, p);
  else if (p.res) TSM_KLAUNCH((conv_bf16_256_kernel<1, false, true, false>), grid, block, kLdsBytes, s, p);
  else if (p.x2) TSM_KLAUNCH((conv_bf16_256_kernel<1, false, false, true>), grid, block, kLdsBytes, s, p);
  else TSM_KLAUNCH((conv_bf16_256_kernel<1, false>), grid, block, kLdsBytes, s, p);
  return hipGetLastError();
}

hipError_t launch_conv_bf16_256p(ConvParams p, int ks, hipStream_t s) {
  if (!conv_bf16_256p_valid(p, ks)) return hipErrorInvalidValue;
  p.ntm = (p.M + 255) / 256;
  p.ntn = p.Cout / 256;
  const DeviceInfo &di = device_info();   // CU count of this device + the > 64 KB dynamic-LDS opt-in
  if (di.status != hipSuccess) return di.status;
  const int ntiles = p.ntm * p.ntn;
  const int slots = di.n_cu & ~7;          // a multiple of 8: a workgroup's tiles then all sit in its own XCD's chunk
  const dim3 grid((unsigned)(ntiles < slots || slots < 8 ? ntiles : slots)), block(512);
  if (ks == 3) TSM_KLAUNCH((conv_bf16_256p_kernel<3, false>), grid, block, kLds256pBytes, s, p);
  else if (p.T > 0) TSM_KLAUNCH((conv_bf16_256p_kernel<1, true>), grid, block, kLds256pBytes, s, p);
  else if (p.res) TSM_KLAUNCH((conv_bf16_256p_kernel<1, false, true, false>), grid, block, kLds256pBytes, s, p);
  else if (p.x2) TSM_KLAUNCH((conv_bf16_256p_kernel<1, false, false, true>), grid, block, kLds256pBytes, s, p);
  else TSM_KLAUNCH((conv_bf16_256p_kernel<1, false>), grid, block, kLds256pBytes, s, p);
  return hipGetLastError();
}

hipError_t opt_in_bf16_256() {
  hipError_t first = hipSuccess;
  auto opt_in = [&](const void *fn, size_t bytes) {
    const hipError_t st = lds_opt_in(fn, bytes);
    if (st != hipSuccess && first == hipSuccess) first = st;
  };
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, false>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, true>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<3, false>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, false, true, false>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, false, false, true>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, false>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, true>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<3, false>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, false, true, false>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, false, false, true>), kLds256pBytes);
  return first;
}

}  // namespace tsm
