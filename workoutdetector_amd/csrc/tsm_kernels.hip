// Hand-written CDNA4 (gfx950) kernels for the TSM-ResNet50 clip forward.
//
//   conv_igemm       implicit-GEMM convolution (1x1 / 3x3 / 7x7, stride 1|2), NHWC activations, LDS-staged A
//                    (im2col rows built on the fly, temporal shift fused into the loader) and B (packed
//                    weights), epilogue = folded-BN bias + residual + ReLU.  Template axes: tile shape and wave
//                    layout (128x128 on 4 or 8 waves, 128x64, 64x64, 32x32 on one wave), KS, SHIFT (fused
//                    temporal shift), RES (residual prefetched under the K loop), PREC (exact-fp32 MFMA /
//                    split-bf16 x3 / bf16), DUAL (second A source concatenated along K = conv3 + downsample in
//                    one GEMM), SEG (fp32 long-K layers: K summed in fixed segments, which makes whole-K and split-K
//                    launches of a layer bit-identical; splitk_reduce adds the segment sums in order).
//                    Every variant accumulates each output in the same k order: results are bit-identical
//                    across tile shapes, pipelines and launch forms of one precision.
//   pack_input       [N,3,H,W] or [N,H,W,3] fp32 -> the stem's input format (fp32: one 4-channel group per pixel;
//                    bf16 formats: one 8-element group per pixel pair)
//   preprocess       fused test transform: uint8/fp32 frames -> resize 256 / crop 224 / normalise -> packed input
//   maxpool3x3s2     NHWC, any storage format
//   temporal_shift   stand-alone NHWC fp32 shift (tests; the forward uses the fused loader)
//   head             per-frame global avg-pool, then mean over segments + FC
//
// Reference semantics: workoutdetector/models/tsm.py:35-50 (shift), :409-419 (forward/head);
// torchvision-0.13 ResNet-50 v1.5 Bottleneck for the conv stack.
#include "tsm_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace tsm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS row stride in floats: 32 data + 4 pad.  With ds_read_b128 (16-lane groups, 64 banks) the 16
// rows of a group land on 16 distinct 4-bank slots (row*36 mod 64 is a permutation of multiples
// of 4), so fragment reads are conflict-free; ds_write_b128 of 8 consecutive lanes covers one row.
constexpr int kLds = 36;
constexpr int kBK = 32;

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution.  GEMM view: Y[M, Cout] = A[M, K] * W^T[K, Cout],
//   M = N*Ho*Wo output pixels, K = KS*KS*C ordered (ky, kx, c) so that NHWC input rows are
//   contiguous along K within one tap.  One workgroup = WGM x WGN waves (4, 8 or 1), block tile
//   BM x BN, K-step = one 128-byte row per operand (32 fp32 or split channels, 64 bf16 channels).
//   Each wave owns (BM/WGM) x (BN/WGN) as TM x TN MFMA tiles of 32x32.
//
// MFMA operand order: v_mfma_f32_32x32x2_f32 takes A[i = lane&31][k = lane>>5] and
// B[k = lane>>5][j = lane&31].  The reduction order inside K is free as long as A and B agree, so
// each lane fetches FOUR consecutive k (one ds_read_b128) at k = 8*kk + 4*(lane>>5) + s and step s
// of the group multiplies element s: per 8 k, one b128 per operand tile feeds 4 MFMAs.
//
// Loader: branch-free.  Both operands come through raw buffer loads whose descriptor is rebased per
// workgroup (so 32-bit byte offsets always suffice) and whose hardware range check returns zeros
// for an offset of kInvalid: im2col padding, rows past M and the zero frames of the temporal shift
// cost a v_cndmask on the offset instead of a branch.  Everything that depends only on the row
// (pixel decode, padding mask, shift validity) is computed once per thread before the K loop.
//
// Epilogue: accumulators go through LDS once so that global traffic is 16 B per lane along Cout
// (residual read, bias, ReLU, store), i.e. whole 128-B lines instead of 4-B scalars.
// ---------------------------------------------------------------------------------------------
constexpr unsigned kInvalid = 0x80000000u;  // >= num_records of every descriptor below

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// split-bf16 helpers.  A 32-byte group is [hi x8 | lo x8]; word w of a half holds elements 2w (low 16
// bits) and 2w+1 (high 16 bits).
__device__ __forceinline__ float split_elem(u32x4 half8, int e) {
  const unsigned w = half8[e >> 1];
  return __builtin_bit_cast(float, (e & 1) ? (w & 0xFFFF0000u) : (w << 16));
}
__device__ __forceinline__ unsigned pack_bf16(float x0, float x1) {  // element 0 in the low half
  const bf16x2 h = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned *hi, unsigned *lo) {
  const bf16x2 h = {(__bf16)x0, (__bf16)x1};  // v_cvt_pk_bf16_f32, round to nearest even
  const unsigned hw = __builtin_bit_cast(unsigned, h);
  const float r0 = x0 - __builtin_bit_cast(float, hw << 16);
  const float r1 = x1 - __builtin_bit_cast(float, hw & 0xFFFF0000u);
  const bf16x2 l = {(__bf16)r0, (__bf16)r1};
  *hi = hw;
  *lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

// PREC selects the arithmetic:
//   kPrecF32     activations/weights fp32, v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain)
//   kPrecBf16x3  "split-bf16": every value is stored as hi = bf16(x), lo = bf16(x - hi), 8 channels per
//                32-byte group [hi x8 | lo x8] (same 4 bytes per element, same byte offsets as fp32, so the
//                loader is shared).  a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16 with fp32
//                accumulation: relative error ~2^-17 per product, three MFMAs at 16x the fp32-MFMA rate.
template <int BM, int BN, int WGM, int WGN, int KS, bool SHIFT, bool RES, int PREC, bool DUAL = false, bool SEG = false>
// (second launch-bounds argument = minimum waves per SIMD: the SEG 64x64 kernel needs 16 registers more than the
// plain one and would drop from 5 to 4 workgroups per CU; asking for 5 costs 1-2 spills outside the K loop)
__global__ void __launch_bounds__(64 * WGM * WGN, (SEG && BM == 64) ? 5 : 1) conv_igemm(const ConvParams p) {
  static_assert(!SEG || (PREC == kPrecF32 && !RES && WGM * WGN <= 4 && BM == BN && BM <= 64),
                "segmented K accumulation: fp32, 64x64 / 32x32 tiles, no residual (ConvParams::kseg_len)");
  static_assert(WGM * WGN == 4 || WGM * WGN == 1 || WGM * WGN == 8,
                "4 waves per workgroup, 1 (32x32 small-M tiles) or 8 (128x128 with 4 waves per SIMD at 2 workgroups/CU)");
  constexpr int NT = 64 * WGM * WGN;   // threads per workgroup
  constexpr int LRP = NT / 8;          // loader rows per pass (8 threads x 16 bytes per 128-byte row)
  static_assert(!DUAL || (KS == 1 && !SHIFT && !RES), "K-concatenated second source: plain 1x1 convs only");
  constexpr bool X3 = PREC == kPrecBf16x3;
  constexpr bool BF = PREC == kPrecBf16;   // plain bf16 storage, one bf16 MFMA per product (config 5)
  constexpr int EB = BF ? 2 : 4;           // bytes per stored element
  constexpr int KC = 128 / EB;             // channels per K-step (an LDS row is always 128 bytes)
  static_assert(!SHIFT || KS == 1, "the temporal shift is fused into 1x1 convs only");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int APASS = BM / LRP, BPASS = BN / LRP;
  // RK ("register-resident K-step", fp32 64x64 / 32x32 tiles): ONE LDS buffer; after the barrier that makes a
  // tile visible every wave pulls all four k-groups of fragments into registers, a second barrier frees the
  // buffer, and the 16 MFMAs of the step then run from registers while the next tile is written into LDS.
  // Half the LDS per workgroup -> more workgroups per CU, and no LDS wait inside the MFMA sequence.
  constexpr bool RK = PREC == kPrecF32 && ((BM == 64 && BN == 64) || (BM == 32 && BN == 32));
  constexpr int NBUF = RK ? 1 : 2;
  constexpr int CLD = BN + 4;  // epilogue staging row stride (floats)
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
  constexpr int SMEM_FLOATS = NBUF * (BM + BN) * kLds > BM * CLD ? NBUF * (BM + BN) * kLds : BM * CLD;

  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l31 = lane & 31;

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of tiles, n fastest, so co-resident blocks re-use the same A panel from L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  if (p.reverse) tile = nwg - 1 - tile;   // walk the tiles from the far end (ConvParams::reverse)
  int seg = 0;  // SEG + ksplit: the grid holds ntm * ntn tiles per K segment, segment-major
  if (SEG && p.ksplit) {
    seg = tile / (p.ntm * p.ntn);
    tile -= seg * (p.ntm * p.ntn);
  }
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- descriptors, rebased to this workgroup's first input frame / first weight row ------------
  const int HoWo = p.Ho * p.Wo;
  const int n_first = m0 / HoWo;
  const int frame0 = SHIFT ? (n_first > 0 ? n_first - 1 : 0) : n_first;
  // bf16-format stem: the input is stored as pixel PAIRS (8-element groups = 2 pixels x 4 channels, odd
  // widths padded with a zero pixel), so a frame is Hi x ceil(Wi/2) groups.
  constexpr bool PAIRS = KS == 7 && PREC != kPrecF32;
  const int wpairs = (p.Wi + 1) >> 1;
  const size_t frame_elems = PAIRS ? (size_t)p.Hi * wpairs * 8 : (size_t)p.Hi * p.Wi * p.C;
  const size_t a_bytes = ((size_t)p.N - frame0) * frame_elems * EB;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)frame0 * frame_elems * EB), 0,
      (int)(a_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(p.w) + (size_t)n0 * p.Kp * EB), 0, BN * p.Kp * EB,
      0x00020000);

  // second A source (DUAL): same output pixels, its own channel count / spatial size / stride
  const size_t frame_elems2 = DUAL ? (size_t)p.Hi2 * p.Wi2 * p.C2 : 0;
  const size_t a2_bytes = DUAL ? ((size_t)p.N - n_first) * frame_elems2 * EB : 0;
  const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(DUAL ? p.x2 : p.x) + (size_t)n_first * frame_elems2 * EB), 0,
      (int)(a2_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a2_bytes), 0x00020000);

  // ---- per-thread loader state: 8 threads per 32-float row, 32 rows per pass -------------------
  const int lrow = tid >> 3;
  const int chunk = tid & 7;
  const int frame_bytes = (int)(frame_elems * EB);
  unsigned a_off[APASS];                       // byte offset of (row, tap 0, this thread's chunk)
  unsigned a_offp[SHIFT ? APASS : 1], a_offm[SHIFT ? APASS : 1];
  unsigned a_mask[KS == 3 ? APASS : 1];
  unsigned a_off2[DUAL ? APASS : 1];
  int a_iy[KS == 7 ? APASS : 1], a_ix[KS == 7 ? APASS : 1];
#pragma unroll
  for (int pp = 0; pp < APASS; ++pp) {
    const int m = m0 + lrow + LRP * pp;
    const bool ok = m < p.M;
    const int mm = ok ? m : m0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    if (KS == 7) {
      a_off[pp] = ok ? (unsigned)((n - frame0) * frame_bytes) : kInvalid;
      a_iy[pp] = iy0;
      // PAIRS: the 7 taps of a row (pixels 2ox-3 .. 2ox+3) sit in the 4 aligned pixel pairs starting at
      // pair ox-2 (pixel 2ox-4, whose weight is zero)
      a_ix[pp] = PAIRS ? (ix0 - 1) >> 1 : ix0;
    } else {
      const int base = (n - frame0) * frame_bytes + (iy0 * p.Wi + ix0) * p.C * EB + chunk * 16;
      a_off[pp] = (KS == 1 && !ok) ? kInvalid : (unsigned)base;
      if (DUAL)
        a_off2[pp] = ok ? (unsigned)((n - n_first) * (int)(frame_elems2 * EB) +
                                     (oy * p.stride2 * p.Wi2 + ox * p.stride2) * p.C2 * EB + chunk * 16)
                        : kInvalid;
      if (KS == 3) {
        unsigned mask = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if ((unsigned)(iy0 + ky) < (unsigned)p.Hi && (unsigned)(ix0 + kx) < (unsigned)p.Wi)
              mask |= 1u << (ky * 3 + kx);
        a_mask[pp] = ok ? mask : 0u;
      }
      if (SHIFT) {
        const int t = n % p.T;  // channels [0,fold) <- frame t+1, [fold,2fold) <- frame t-1
        a_offp[pp] = (ok && t < p.T - 1) ? (unsigned)(base + frame_bytes) : kInvalid;
        a_offm[pp] = (ok && t > 0) ? (unsigned)(base - frame_bytes) : kInvalid;
      }
    }
  }
  unsigned b_off[BPASS];
#pragma unroll
  for (int pp = 0; pp < BPASS; ++pp) b_off[pp] = (unsigned)((lrow + LRP * pp) * p.Kp * EB + chunk * 16);

  f32x4 ra[APASS], rb[BPASS];
  const int nk1 = DUAL ? p.K1 / KC : 0;

  // Loader work is cut into NITEMS = APASS + BPASS single-instruction items (one 16-B buffer load or
  // one ds_write_b128 each) so the main loop can drop one item between consecutive MFMAs.
  // A dead K-step (past the end of K) ORs kInvalid into every offset, so its loads return zeros
  // without touching memory and the loop body stays branch-free.
  constexpr int NITEMS = APASS + BPASS;
  struct KStep {  // wave-uniform per-K-step scalars
    unsigned kbytes;
    int tap, tap_off;
    unsigned dead;  // 0 for a live step, kInvalid for a step past the end of K (pure arithmetic, no select)
    unsigned mp, mm, m0;  // SHIFT: lane masks choosing the t+1 / t-1 / t source frame for this thread's chunk
  };
  auto kstep = [&](int kt, int nk_) {
    KStep k;
    k.kbytes = (unsigned)kt * (kBK * 4);
    k.dead = (~(unsigned)((kt - nk_) >> 31)) & kInvalid;
    k.mp = k.mm = 0u;
    k.m0 = ~0u;
    if (SHIFT) {
      // channels [0,fold) <- frame t+1, [fold,2fold) <- frame t-1, rest <- frame t.  Kept as AND/OR
      // masks: a three-way select over the per-row offset arrays is turned into a scratch-memory
      // table by the compiler, which serialises the loader behind vmcnt(0).
      const int c = kt * KC + (X3 ? (chunk >> 1) * 8 : (BF ? chunk * 8 : chunk * 4));
      k.mp = 0u - (unsigned)(c < p.fold);
      k.mm = (0u - (unsigned)(c < 2 * p.fold)) & ~k.mp;
      k.m0 = ~(k.mp | k.mm);
    }
    k.tap = 0;
    k.tap_off = 0;
    if (KS == 3) {  // C >= 32 so a K-step never straddles a tap: tap and its offset are scalars
      k.tap = (kt * KC) >> (p.logC4 + 2);
      const int c0 = kt * KC - k.tap * p.C;
      const int ky = k.tap / 3, kx = k.tap - ky * 3;
      k.tap_off = ((ky * p.Wi + kx) * p.C + c0) * EB;
    }
    return k;
  };
  auto gload_item = [&](const KStep &k, int kt, int item) {
    if (item < APASS) {
      const int pp = item;
      if (KS == 1) {
        unsigned off = a_off[pp];
        if (SHIFT) off = (a_offp[pp] & k.mp) | (a_offm[pp] & k.mm) | (a_off[pp] & k.m0);
        if (DUAL) {
          // K-steps [0, nk1) come from the first source, the rest from the second (wave-uniform choice)
          const bool second = kt >= nk1;
          ra[pp] = buf_load4(second ? rsrcA2 : rsrcA, (second ? a_off2[pp] : off) | k.dead,
                             second ? k.kbytes - (unsigned)nk1 * 128u : k.kbytes);
        } else {
          ra[pp] = buf_load4(rsrcA, off | k.dead, k.kbytes);
        }
      } else if (KS == 3) {
        ra[pp] = buf_load4(rsrcA, (((a_mask[pp] >> k.tap) & 1u) ? a_off[pp] + (unsigned)k.tap_off : kInvalid) | k.dead, 0);
      } else {
        if constexpr (PAIRS) {
          // bf16-format stem: K = (ky, pair j, pixel-in-pair, c4) = 7 x 4 x 8 = 224; one 8-element group
          // per 16-B chunk (bf16) or per chunk pair hi/lo (split); groups >= 28 are K padding
          const int g = X3 ? (kt * 8 + chunk) >> 1 : kt * 8 + chunk;
          const int ky = g >> 2, j = g & 3;
          const int iy = a_iy[pp] + ky, pc = a_ix[pp] + j;
          const bool ok = g < 28 && (unsigned)iy < (unsigned)p.Hi && (unsigned)pc < (unsigned)wpairs;
          const unsigned pix = X3 ? (unsigned)((iy * wpairs + pc) * 32 + (chunk & 1) * 16)
                                  : (unsigned)((iy * wpairs + pc) * 16);
          ra[pp] = buf_load4(rsrcA, (ok ? a_off[pp] + pix : kInvalid) | k.dead, 0);
        } else {
          // fp32 stem: C = 4 -> one tap per 16-B chunk; taps >= 49 are K padding
          const int tap = kt * 8 + chunk;
          const int ky = tap / 7, kx = tap - ky * 7;
          const int iy = a_iy[pp] + ky, ix = a_ix[pp] + kx;
          const bool ok = tap < 49 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
          ra[pp] = buf_load4(rsrcA, (ok ? a_off[pp] + (unsigned)((iy * p.Wi + ix) * 16) : kInvalid) | k.dead, 0);
        }
      }
    } else {
      const int pp = item - APASS;
      rb[pp] = buf_load4(rsrcB, b_off[pp] | k.dead, k.kbytes);
    }
  };
  auto lstore_item = [&](int buf, int item) {
    float *As = smem + buf * (BM + BN) * kLds;
    if (item < APASS)
      *reinterpret_cast<f32x4 *>(As + (lrow + LRP * item) * kLds + chunk * 4) = ra[item];
    else
      *reinterpret_cast<f32x4 *>(As + BM * kLds + (lrow + LRP * (item - APASS)) * kLds + chunk * 4) = rb[item - APASS];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Fragment registers are double-buffered across the four 8-deep k-groups of a K-step so that the
  // LDS latency of group kk+1 hides under the 16 MFMAs (1024 cycles) of group kk.
  f32x4 af[2][TM], bf[2][TN];
  auto frag_load = [&](int buf, int kk, int set) {
    const float *As = smem + buf * (BM + BN) * kLds + (wm * WTM + l31) * kLds + half * 4;
    const float *Bs = smem + buf * (BM + BN) * kLds + BM * kLds + (wn * WTN + l31) * kLds + half * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const f32x4 *>(As + i * 32 * kLds + kk * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const f32x4 *>(Bs + j * 32 * kLds + kk * 8);
  };
  // 4*TM*TN MFMAs of one k-group; `inject(item)` is called NITEMS times, spread evenly between them,
  // and a scheduling fence pins each injected instruction behind the MFMA it follows: the matrix
  // pipe executes an issued MFMA for 64 cycles, during which the wave may issue the injected item.
  constexpr int NMFMA = 4 * TM * TN;
  auto mfma_group = [&](int set, auto &&inject) {
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][s], bf[set][j][s], acc[i][j], 0, 0, 0);
          ++cnt;
          const int done = (cnt * NITEMS) / NMFMA, before = ((cnt - 1) * NITEMS) / NMFMA;
#pragma unroll
          for (int it = before; it < done; ++it) {
            inject(it);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };
  auto no_inject = [](int) {};
  auto mfma_plain = [&](int set) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][s], bf[set][j][s], acc[i][j], 0, 0, 0);
  };
  (void)no_inject;

  // ---- split-bf16 fragments: an LDS row is 4 channel groups of [hi x8 | lo x8]; v_mfma_f32_32x32x16_bf16
  // takes A[row][k = 8*(lane>>5) + j], so k16-group q reads channel group g = 2q + (lane>>5).
  u32x4 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  auto frag_load_x3 = [&](int buf, int qg, int set) {
    const float *As = smem + buf * (BM + BN) * kLds + (wm * WTM + l31) * kLds + (2 * qg + half) * 8;
    const float *Bs = smem + buf * (BM + BN) * kLds + BM * kLds + (wn * WTN + l31) * kLds + (2 * qg + half) * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[set][i] = *reinterpret_cast<const u32x4 *>(As + i * 32 * kLds);
      al[set][i] = *reinterpret_cast<const u32x4 *>(As + i * 32 * kLds + 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[set][j] = *reinterpret_cast<const u32x4 *>(Bs + j * 32 * kLds);
      bl[set][j] = *reinterpret_cast<const u32x4 *>(Bs + j * 32 * kLds + 4);
    }
  };
  constexpr int NMFMA3 = 3 * TM * TN;
  auto mfma_x3 = [&](int set, int nitems, auto &&inject) {
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, t == 2 ? al[set][i] : ah[set][i]);
          const bf16x8 b = __builtin_bit_cast(bf16x8, t == 1 ? bl[set][j] : bh[set][j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
          ++cnt;
          const int done = (cnt * nitems) / NMFMA3, before = ((cnt - 1) * nitems) / NMFMA3;
#pragma unroll
          for (int it = before; it < done; ++it) {
            inject(it);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };

  // ---- plain bf16: a 128-byte LDS row is 64 channels = four k16-groups; group q, lane half h reads the 8
  // channels 16q + 8h (one ds_read_b128 per operand tile per MFMA).
  u32x4 af16[4][TM], bf16f[4][TN];
  auto frag_load_bf = [&](int buf, int qg) {
    const float *As = smem + buf * (BM + BN) * kLds + (wm * WTM + l31) * kLds + qg * 8 + half * 4;
    const float *Bs = smem + buf * (BM + BN) * kLds + BM * kLds + (wn * WTN + l31) * kLds + qg * 8 + half * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) af16[qg][i] = *reinterpret_cast<const u32x4 *>(As + i * 32 * kLds);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf16f[qg][j] = *reinterpret_cast<const u32x4 *>(Bs + j * 32 * kLds);
  };
  constexpr int NMFMA_BF = TM * TN;
  auto mfma_bf = [&](int qg, int item0, int nitems, auto &&inject) {
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af16[qg][i]),
                                                            __builtin_bit_cast(bf16x8, bf16f[qg][j]), acc[i][j], 0, 0, 0);
        ++cnt;
        const int done = (cnt * nitems) / NMFMA_BF, before = ((cnt - 1) * nitems) / NMFMA_BF;
#pragma unroll
        for (int it = before; it < done; ++it) {
          inject(item0 + it);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };

  // Residual tile: fetched before the K loop (it does not depend on it) in the epilogue's own
  // row-major 16-B mapping, so its HBM latency hides under the MFMAs.  Rows past M read as zeros.
  constexpr int EW = (X3 || BF) ? 8 : 4;  // channels per thread per pass (split: one 32-byte group; bf16: 16 bytes)
  constexpr int TPR = BN / EW;         // threads per output row
  constexpr int RPP = NT / TPR;        // rows per pass
  constexpr int EPASS = BM / RPP;
  const int ecol = (tid % TPR) * EW, erow = tid / TPR;
  f32x4 rres[(RES && !X3 && !BF) ? EPASS : 1];
  u32x4 rres_h[(RES && (X3 || BF)) ? EPASS : 1], rres_l[(RES && X3) ? EPASS : 1];
  if (RES && (X3 || BF)) {
    const size_t r_bytes = ((size_t)p.M - m0) * p.Cout * EB;
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.res) + (size_t)m0 * p.Cout * EB), 0,
        (int)(r_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : r_bytes), 0x00020000);
#pragma unroll
    for (int k = 0; k < EPASS; ++k) {
      const unsigned o = (unsigned)(((erow + k * RPP) * p.Cout + n0 + ecol) * EB);
      rres_h[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)o, 0, 0);
      if (X3) rres_l[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)(o + 16), 0, 0);
    }
  }
  if (RES && !X3 && !BF) {
    const size_t r_bytes = ((size_t)p.M - m0) * p.Cout * 4;
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.res + (size_t)m0 * p.Cout), 0,
        (int)(r_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : r_bytes), 0x00020000);
#pragma unroll
    for (int k = 0; k < EPASS; ++k)
      rres[k] = buf_load4(rsrcR, (unsigned)(((erow + k * RPP) * p.Cout + n0 + ecol) * 4), 0);
  }

  // ---- main loop -------------------------------------------------------------------------------
  // Two LDS buffers, one barrier per K-step.  Iteration kt multiplies tile kt out of buf[kt&1] in four
  // k-groups of 4*TM*TN MFMAs:
  //   group 0            plain
  //   group 1            + ds_writes of tile kt+1 (registers, loaded during iteration kt-1) into the
  //                        other buffer: its last readers finished before the barrier of kt-1
  //   group 2            + buffer loads of tile kt+2, which then have a whole K-step to land
  //   barrier
  //   group 3            register-only, issued AFTER the barrier so that it covers the LDS latency of
  //                        the next tile's first fragments (read right after the barrier)
  // One barrier per step suffices: tile kt+1 is complete in LDS before it, and nobody overwrites
  // buf[kt&1] before the next barrier.  The body is straight-line.
  // This workgroup multiplies K-steps [kt0, nk): all of K, or one segment of it (SEG + ksplit).
  int kt0 = 0, nk = p.Kp / KC;
  if (SEG && p.ksplit) {
    kt0 = seg * p.kseg_len;
    nk = kt0 + p.kseg_len < nk ? kt0 + p.kseg_len : nk;
  }
  // SEG: `acc` holds the running segment, `tot` the sum of the finished ones (segment boundaries sit at
  // multiples of kseg_len from K-step 0 in both launch forms; kt0 is such a multiple).
  f32x16 tot[SEG ? TM : 1][SEG ? TN : 1];
  if constexpr (SEG) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[i][j][e] = 0.f;
  }
  auto seg_flush = [&]() {
    if constexpr (SEG) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
    }
  };
  // SEG loops are nested: the inner loop runs the K-steps of one segment, the flush sits between segments (kept
  // out of the inner loop on purpose: inside it the compiler if-converts the flush and drains the MFMA chain on
  // every K-step).  Without SEG there is a single pass over [kt0, nk).
  const int seg_len = (SEG && p.kseg_len > 0) ? p.kseg_len : 0x3fffffff;
  {
    const KStep k0 = kstep(kt0, nk);
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) gload_item(k0, kt0, it);
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) lstore_item(0, it);
    const KStep k1 = kstep(kt0 + 1, nk);
#pragma unroll
    for (int it = 0; it < NITEMS; ++it) gload_item(k1, kt0 + 1, it);
  }
  __syncthreads();
  if constexpr (BF) {
    // plain bf16: four k16-groups of TM*TN MFMAs per K-step.  Groups 0-1 carry the ds_writes of tile kt+1
    // and the buffer loads of tile kt+2; groups 2-3 run after the barrier and cover the next fragments.
    frag_load_bf(0, 0);
    frag_load_bf(0, 1);
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      const KStep k2 = kstep(kt + 2, nk);
      frag_load_bf(cur, 2);
      frag_load_bf(cur, 3);
      auto inject = [&](int it) {
        if (it < NITEMS) lstore_item(cur ^ 1, it);
        else gload_item(k2, kt + 2, it - NITEMS);
      };
      mfma_bf(0, 0, NITEMS, inject);
      mfma_bf(1, NITEMS, NITEMS, inject);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      mfma_bf(2, 0, 0, [](int) {});
      frag_load_bf(cur ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_bf(3, 0, 0, [](int) {});
      __builtin_amdgcn_sched_barrier(0);
      frag_load_bf(cur ^ 1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (RK) {
    f32x4 ra_[4], rb_[4];  // all four k-groups of this wave's A / B fragments (TM = TN = 1)
    // fp32 stem: K = 49 taps x 4 channels = 196 of the 224 padded, so the last K-step holds real data in its first
    // 4 k only: it runs as a 4-MFMA tail instead of 16 (the 12 skipped MFMAs multiply zeros; same bits).
    const bool trim = KS == 7 && p.Kp == 224 && nk == 7;
    const int nk_full = trim ? nk - 1 : nk;
    for (int kt = kt0; kt < nk_full;) {
    const int kend = (SEG && kt + seg_len < nk_full) ? kt + seg_len : nk_full;
    for (; kt < kend; ++kt) {
      const KStep k2 = kstep(kt + 2, nk);
      {
        const float *As = smem + (wm * WTM + l31) * kLds + half * 4;
        const float *Bs = smem + BM * kLds + (wn * WTN + l31) * kLds + half * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          ra_[kk] = *reinterpret_cast<const f32x4 *>(As + kk * 8);
          rb_[kk] = *reinterpret_cast<const f32x4 *>(Bs + kk * 8);
        }
      }
      __syncthreads();  // every wave holds its fragments: the buffer may be overwritten
      int cnt = 0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_[kk][s4], rb_[kk][s4], acc[0][0], 0, 0, 0);
          ++cnt;
          // 2*NITEMS loader items over the first 12 MFMAs: the ds_writes of tile kt+1, then the loads of kt+2
          const int done = cnt < 12 ? (cnt * 2 * NITEMS) / 12 : 2 * NITEMS;
          const int before = cnt - 1 < 12 ? ((cnt - 1) * 2 * NITEMS) / 12 : 2 * NITEMS;
#pragma unroll
          for (int it = before; it < done; ++it) {
            if (it < NITEMS) lstore_item(0, it);
            else gload_item(k2, kt + 2, it - NITEMS);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      __syncthreads();  // tile kt+1 is complete in LDS
    }
    seg_flush();
    }
    if (KS == 7 && trim) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(smem + (wm * WTM + l31) * kLds + half * 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(smem + BM * kLds + (wn * WTN + l31) * kLds + half * 4);
      __syncthreads();  // fragments are in registers: the epilogue may reuse the buffer
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], b0[s4], acc[0][0], 0, 0, 0);
    }
  } else if constexpr (!X3) {
    frag_load(0, 0, 0);
    frag_load(0, 1, 1);
    for (int kt = kt0; kt < nk;) {
    const int kend = (SEG && kt + seg_len < nk) ? kt + seg_len : nk;
    for (; kt < kend; ++kt) {
      const int cur = (kt - kt0) & 1;
      const KStep k2 = kstep(kt + 2, nk);
      mfma_plain(0);
      frag_load(cur, 2, 0);
      mfma_group(1, [&](int it) { lstore_item(cur ^ 1, it); });
      frag_load(cur, 3, 1);
      mfma_group(0, [&](int it) { gload_item(k2, kt + 2, it); });
      __syncthreads();
      frag_load(cur ^ 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_plain(1);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(cur ^ 1, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    seg_flush();
    }
  } else {
    // split-bf16: a K-step is two k16-groups of 3*TM*TN MFMAs (32 cycles each).  Group 0 carries the
    // ds_writes of tile kt+1 and the buffer loads of tile kt+2; group 1 is issued after the barrier and
    // covers the LDS latency of the next tile's fragments.  Same buffer/barrier reasoning as above.
    frag_load_x3(0, 0, 0);
    frag_load_x3(0, 1, 1);
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      const KStep k2 = kstep(kt + 2, nk);
      mfma_x3(0, 2 * NITEMS, [&](int it) {
        if (it < NITEMS) lstore_item(cur ^ 1, it);
        else gload_item(k2, kt + 2, it - NITEMS);
      });
      __syncthreads();
      frag_load_x3(cur ^ 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_x3(1, 0, [](int) {});
      __builtin_amdgcn_sched_barrier(0);
      frag_load_x3(cur ^ 1, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Stage the BM x BN tile in
  // LDS (all operand reads finished at the barrier above), then stream it out row-major.
  float *Cs = smem;  // (SEG: every segment, the last one included, was flushed into tot after its inner loop)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[(wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * CLD + wn * WTN + j * 32 + l31] =
            SEG ? tot[SEG ? i : 0][SEG ? j : 0][e] : acc[i][j][e];
  __syncthreads();

  // Stores go through a descriptor that ends at row M: rows past the end are dropped by the range
  // check, which keeps the epilogue branch-free (no per-pass wait on earlier stores).
  const bool partial = SEG && p.ksplit;  // raw segment sums to y = partial[seg][M][Cout]: no bias, no ReLU
  const size_t y_bytes = ((size_t)p.M - m0) * p.Cout * EB;
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char *>(p.y) + ((size_t)(partial ? seg : 0) * p.M + m0) * p.Cout * EB, 0,
      (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes), 0x00020000);
  const float floor_ = (p.relu && !partial) ? 0.f : -INFINITY;  // ReLU as a branch-free clamp
  if constexpr (BF) {
    const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + ecol);
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + ecol + 4);
#pragma unroll
    for (int k = 0; k < EPASS; ++k) {
      const int rr = erow + k * RPP;
      const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cs + rr * CLD + ecol);
      const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cs + rr * CLD + ecol + 4);
      float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                    c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
      if (RES) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += split_elem(rres_h[k], e);
      }
      u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) o[w] = pack_bf16(fmaxf(v[2 * w], floor_), fmaxf(v[2 * w + 1], floor_));
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (rr * p.Cout + n0 + ecol) * 2, 0, 0);
    }
  } else if constexpr (!X3) {
    f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + n0 + ecol);
    if (partial) bias = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < EPASS; ++k) {
      const int rr = erow + k * RPP;
      f32x4 v = *reinterpret_cast<const f32x4 *>(Cs + rr * CLD + ecol);
      v += bias;
      if (RES) v += rres[k];
      v[0] = fmaxf(v[0], floor_);
      v[1] = fmaxf(v[1], floor_);
      v[2] = fmaxf(v[2], floor_);
      v[3] = fmaxf(v[3], floor_);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcY,
                                             (int)((rr * p.Cout + n0 + ecol) * 4), 0, 0);
    }
  } else {
    const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + ecol);
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + ecol + 4);
#pragma unroll
    for (int k = 0; k < EPASS; ++k) {
      const int rr = erow + k * RPP;
      const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cs + rr * CLD + ecol);
      const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cs + rr * CLD + ecol + 4);
      float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                    c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
      if (RES) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += split_elem(rres_h[k], e) + split_elem(rres_l[k], e);
      }
      u32x4 oh, ol;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float x0 = fmaxf(v[2 * w], floor_), x1 = fmaxf(v[2 * w + 1], floor_);
        unsigned hw, lw;
        split_pair(x0, x1, &hw, &lw);
        oh[w] = hw;
        ol[w] = lw;
      }
      const int o = (rr * p.Cout + n0 + ecol) * 4;
      __builtin_amdgcn_raw_buffer_store_b128(oh, rsrcY, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(ol, rsrcY, o + 16, 0, 0);
    }
  }
}

int conv_num_segments(const ConvParams &p) {
  if (p.kseg_len <= 0) return 1;
  const int nk = p.Kp / kBK;
  return (nk + p.kseg_len - 1) / p.kseg_len;
}

// Segmented-K instantiations (fp32, 64x64 / 32x32 tiles, no residual): one workgroup per tile, or per
// (tile, segment) when p.ksplit is set.
template <int BM, int BN, int WGM, int WGN, int KS, bool SHIFT>
static hipError_t launch_conv_seg(ConvParams p, hipStream_t s) {
  if constexpr (!((BM == 64 && BN == 64) || (BM == 32 && BN == 32))) {
    return hipErrorInvalidValue;
  } else {
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = p.Cout / BN;
    const dim3 grid((unsigned)(p.ntm * p.ntn * (p.ksplit ? conv_num_segments(p) : 1)));
    const dim3 block(64 * WGM * WGN);
    if constexpr (KS == 1 && !SHIFT) {
      if (p.x2) {
        hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecF32, true, true>), grid, block, 0, s, p);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, false, kPrecF32, false, true>), grid, block, 0, s, p);
    return hipGetLastError();
  }
}

template <int BM, int BN, int WGM, int WGN, int KS, bool SHIFT, bool RES>
static hipError_t launch_conv_t(ConvParams p, hipStream_t s) {
  if (p.kseg_len > 0) {
    if constexpr (!RES && KS != 7) return launch_conv_seg<BM, BN, WGM, WGN, KS, SHIFT>(p, s);
    else return hipErrorInvalidValue;
  }
  p.ntm = (p.M + BM - 1) / BM;
  p.ntn = p.Cout / BN;
  const dim3 grid((unsigned)(p.ntm * p.ntn));
  if constexpr (KS == 1 && !SHIFT && !RES) {
    if (p.x2) {
      if (p.prec == kPrecBf16x3)
        hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecBf16x3, true>), grid, dim3(64 * WGM * WGN), 0, s, p);
      else if (p.prec == kPrecBf16)
        hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecBf16, true>), grid, dim3(64 * WGM * WGN), 0, s, p);
      else
        hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecF32, true>), grid, dim3(64 * WGM * WGN), 0, s, p);
      return hipGetLastError();
    }
  }
  if (p.prec == kPrecBf16x3)
    hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecBf16x3>), grid, dim3(64 * WGM * WGN), 0, s, p);
  else if (p.prec == kPrecBf16)
    hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecBf16>), grid, dim3(64 * WGM * WGN), 0, s, p);
  else
    hipLaunchKernelGGL((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecF32>), grid, dim3(64 * WGM * WGN), 0, s, p);
  return hipGetLastError();
}

void conv_tile_shape(const ConvParams &p, int *bm, int *bn) {
  // Cout is a multiple of 64 everywhere in ResNet-50.  Prefer 128x128; fall back to smaller tiles
  // when the grid would leave most of the 256 CUs idle (small M at batch 1).
  int BN = (p.Cout % 128 == 0) ? 128 : 64;
  int BM = 128;
  const long tiles128 = (long)((p.M + 127) / 128) * (p.Cout / BN);
  if (tiles128 < 256) {
    BM = 64;
    BN = 64;
  }
  *bm = BM;
  *bn = BN;
}

int conv_tile_from_name(const char *name) {
  if (!name) return kTileAuto;
  static const struct { const char *n; int t; } names[] = {{"128x128", kTile128x128}, {"128x64", kTile128x64},
      {"64x64", kTile64x64}, {"32x32", kTile32x32}, {"128x128w8", kTile128x128w8}, {"256x256", kTile256x256}, {"ws", kTileWs}, {"256x256p", kTile256x256p}};
  for (const auto &e : names)
    if (strcmp(name, e.n) == 0) return e.t;
  return kTileAuto;
}

bool conv_tile_valid(const ConvParams &p, int tile) {
  switch (tile) {
    case kTile128x128: return p.Cout % 128 == 0;
    case kTile128x64:
    case kTile64x64: return p.Cout % 64 == 0;
    case kTile32x32: return p.Cout % 32 == 0 && p.prec == kPrecF32;  // single-wave tiles: fp32 only
    case kTile128x128w8: return p.Cout % 128 == 0;
    case kTile256x256:   // (ks is checked at launch: the stem has C == 4 and never qualifies)
      return p.prec == kPrecBf16 && p.Cout % 256 == 0 && p.C % 64 == 0 && !(p.res && p.x2) &&
             (!p.x2 || (p.K1 % 64 == 0 && p.C2 % 64 == 0));
    case kTile256x256p:   // (ks is checked at launch; at least two K-tiles, the bias of all channels in LDS)
      return p.prec == kPrecBf16 && p.Cout % 256 == 0 && p.Cout <= 2048 && p.C % 64 == 0 && p.Kp >= 128 && !(p.res && p.x2) &&
             (!p.x2 || (p.K1 % 64 == 0 && p.C2 % 64 == 0));
    case kTileWs: return conv3x3_ws_valid(p) || conv3x3_ws128_valid(p) || conv1x1_ws_valid(p) || conv1x1_wsn_valid(p);   // (pad singles out 3x3 / 1x1)
    default: return false;
  }
}

void conv_tile_dims(int tile, int *bm, int *bn) {
  *bm = (tile == kTile256x256 || tile == kTile256x256p || tile == kTileWs) ? 256 : tile == kTile32x32 ? 32 : (tile == kTile64x64 ? 64 : 128);
  *bn = (tile == kTile256x256 || tile == kTile256x256p) ? 256 : tile == kTile32x32 ? 32 : ((tile == kTile128x128 || tile == kTile128x128w8) ? 128 : 64);
}

static hipError_t launch_conv_bf16_256(ConvParams p, int ks, hipStream_t s);
static hipError_t launch_conv_bf16_256p(ConvParams p, int ks, hipStream_t s);
static hipError_t launch_conv3x3_ws(ConvParams p, hipStream_t s);
static hipError_t launch_conv1x1_ws(const ConvParams &p, hipStream_t s);
static hipError_t launch_conv1x1_wsn(const ConvParams &p, hipStream_t s);

template <int KS, bool SHIFT, bool RES>
static hipError_t launch_conv_ks(const ConvParams &p_in, hipStream_t s) {
  ConvParams p = p_in;
  int bm, bn;
  conv_tile_shape(p, &bm, &bn);
  if (p.tile != kTileAuto) {
    if (!conv_tile_valid(p, p.tile)) return hipErrorInvalidValue;
    conv_tile_dims(p.tile, &bm, &bn);
  }
  if (p.tile == kTile256x256) {
    if constexpr (KS != 7) return launch_conv_bf16_256(p, KS, s);
    else return hipErrorInvalidValue;
  }
  if (p.tile == kTile256x256p) {
    if constexpr (KS != 7) return launch_conv_bf16_256p(p, KS, s);
    else return hipErrorInvalidValue;
  }
  if (p.tile == kTileWs) {
    if constexpr (KS == 3) return launch_conv3x3_ws(p, s);
    else if constexpr (KS == 1 && !RES) return conv1x1_ws_valid(p) ? launch_conv1x1_ws(p, s) : launch_conv1x1_wsn(p, s);
    else return hipErrorInvalidValue;
  }
  if (p.kseg_len > 0 && !(bm == 32 && bn == 32)) {  // segmented accumulation exists on 64x64 / 32x32 tiles only
    bm = 64;
    bn = 64;
    if (p.tile == kTile128x128w8) p.tile = kTile64x64;
  }
  if (bm == 32 && bn == 32) {
    if (p.prec != kPrecF32) return hipErrorInvalidValue;
    return launch_conv_t<32, 32, 1, 1, KS, SHIFT, RES>(p, s);
  }
  if (bm == 128 && bn == 128 && p.tile == kTile128x128w8) return launch_conv_t<128, 128, 4, 2, KS, SHIFT, RES>(p, s);
  if (bm == 128 && bn == 128) return launch_conv_t<128, 128, 2, 2, KS, SHIFT, RES>(p, s);
  if (bm == 128 && bn == 64) return launch_conv_t<128, 64, 2, 2, KS, SHIFT, RES>(p, s);
  return launch_conv_t<64, 64, 2, 2, KS, SHIFT, RES>(p, s);
}

hipError_t launch_conv(const ConvParams &p_in, int ks, hipStream_t s) {
  ConvParams p = p_in;
  const int kc = p.prec == kPrecBf16 ? 64 : kBK;  // channels per K-step
  if (p.Cout % 64 != 0 || p.Kp % kc != 0 || p.M <= 0) return hipErrorInvalidValue;
  if ((1 << p.logC4) * 4 != p.C) return hipErrorInvalidValue;
  if (ks != 7 && p.C % kc != 0) return hipErrorInvalidValue;
  if (p.T > 0 && (ks != 1 || p.stride != 1 || p.N % p.T != 0 || p.fold % 4 != 0)) return hipErrorInvalidValue;
  if (p.x2 && (ks != 1 || p.T > 0 || p.res || p.K1 % kc != 0 || p.C2 % kc != 0 || p.K1 + p.C2 != p.Kp || p.K1 != p.C))
    return hipErrorInvalidValue;
  if (p.prec != kPrecF32 && p.prec != kPrecBf16x3 && p.prec != kPrecBf16) return hipErrorInvalidValue;
  if (p.kseg_len < 0 || (p.kseg_len > 0 && (p.prec != kPrecF32 || p.res || ks == 7))) return hipErrorInvalidValue;
  if (p.ksplit && p.kseg_len <= 0) return hipErrorInvalidValue;
  if (p.prec != kPrecF32 && p.T > 0 && p.fold % 8 != 0) return hipErrorInvalidValue;
  // stem: 4 channels per pixel (3 + a zero); the bf16 formats read pixel pairs, which needs stride 2 / pad 3
  if (ks == 7 && (p.C != 4 || (p.prec != kPrecF32 && (p.stride != 2 || p.pad != 3)))) return hipErrorInvalidValue;
  // 32-bit byte offsets inside a workgroup's rebased window: a tile touches at most
  // BM/(Ho*Wo) + 4 input frames.
  const double frames = 128.0 / ((double)p.Ho * p.Wo) + 4.0;
  if (frames * (double)p.Hi * p.Wi * p.C * 4.0 > 2.0e9) return hipErrorInvalidValue;
  switch (ks) {
    case 1:
      if (p.res) return p.T > 0 ? hipErrorInvalidValue : launch_conv_ks<1, false, true>(p, s);
      return p.T > 0 ? launch_conv_ks<1, true, false>(p, s) : launch_conv_ks<1, false, false>(p, s);
    case 3: return p.res ? hipErrorInvalidValue : launch_conv_ks<3, false, false>(p, s);
    case 7: return p.res ? hipErrorInvalidValue : launch_conv_ks<7, false, false>(p, s);
    default: return hipErrorInvalidValue;
  }
}

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
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (row * p.Cout + n0 + wn * 64 + c8 * 8) * 2, 0, 0);
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
  auto stage_of = [&](const State &T, int kt, unsigned par, int which, unsigned dead) {
    const int kh = which >> 1;
    const unsigned kbytes = (unsigned)kt * 128u + (unsigned)kh * 64u;
    unsigned char *dst = lds + par * 65536u + ((which & 1) * 2 + kh) * 16384 + wave * 2048;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(T.pa), 0, T.sza, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(T.pb), 0, 256 * p.Kp * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(T.pa2), 0, T.sza2, 0x00020000);
    if (which & 1) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
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
    setup(nxt, s + 1);
    for (int kt = 0; kt < nt; ++kt, ++g) {
      const unsigned buf = (unsigned)(g & 1) * 65536u;
      // K-tile kt + d of the flat sequence: this tile's, or the first ones of the next tile
      auto stage = [&](int d, int which) {
        const unsigned par = (unsigned)((g + d) & 1);
        if (kt + d < nt) stage_of(cur, kt + d, par, which, 0u);
        else stage_of(nxt, kt + d - nt, par, which, next_dead);
      };
      const bool after_epilogue = kt == 0 && s > 0;       // 16 stores sit between the operands awaited here and the younger DMA
      const bool with_res = RES && kt == nt - 1;          // 8 residual loads sit there (issued right here)
      if (with_res) {
#pragma unroll
        for (int t = 0; t < 8; ++t) load_res(cur, t, t);
      }
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
        if (ph == 0) stage(1, 2);
        else if (ph == 1) stage(1, 3);
        else if (ph == 2) stage(2, 0);
        else stage(2, 1);
        if (ph & 1) {
          if (after_epilogue) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
          else if (with_res) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = RES ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
                                                                      acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af[i]),
                                                                      acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
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
            __builtin_amdgcn_raw_buffer_store_b128(ov, rsrcY, o, j * 64 + qq * 16, 0);
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
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, ((wm * 128 + t * 8 + r8l) * p.Cout + cur.n0 + wn * 64 + c8 * 8) * 2, 0, 0);
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
    cur = nxt;
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();              // the early group waits for the delayed one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dead tail stages (zeros) land before the workgroup leaves its LDS
}

bool conv_bf16_256p_valid(const ConvParams &p, int ks) {
  return conv_bf16_256_valid(p, ks) && p.Kp >= 128 && p.Cout <= 2048;
}

// Per DEVICE, once: the CU count that sizes the persistent grids and the > 64 KB dynamic-LDS opt-in of every kernel
// that needs one.  tsm_hip.h lets engines on several devices live in one process, so neither may be cached from
// whichever device happened to launch first (defined below the last kernel it names).
struct DeviceInfo {
  int n_cu = 256;
  hipError_t status = hipSuccess;
};
static const DeviceInfo &device_info();
constexpr size_t kLds256Bytes = 131072;

static hipError_t launch_conv_bf16_256(ConvParams p, int ks, hipStream_t s) {
  if (!conv_bf16_256_valid(p, ks)) return hipErrorInvalidValue;
  p.ntm = (p.M + 255) / 256;
  p.ntn = p.Cout / 256;
  const dim3 grid((unsigned)(p.ntm * p.ntn)), block(512);
  constexpr size_t kLdsBytes = kLds256Bytes;
  const DeviceInfo &di = device_info();   // the > 64 KB dynamic-LDS opt-in, once per device
  if (di.status != hipSuccess) return di.status;
  if (ks == 3) hipLaunchKernelGGL((conv_bf16_256_kernel<3, false>), grid, block, kLdsBytes, s, p);
  else if (p.T > 0) hipLaunchKernelGGL((conv_bf16_256_kernel<1, true>), grid, block, kLdsBytes, s, p);
  else if (p.res) hipLaunchKernelGGL((conv_bf16_256_kernel<1, false, true, false>), grid, block, kLdsBytes, s, p);
  else if (p.x2) hipLaunchKernelGGL((conv_bf16_256_kernel<1, false, false, true>), grid, block, kLdsBytes, s, p);
  else hipLaunchKernelGGL((conv_bf16_256_kernel<1, false>), grid, block, kLdsBytes, s, p);
  return hipGetLastError();
}

static hipError_t launch_conv_bf16_256p(ConvParams p, int ks, hipStream_t s) {
  if (!conv_bf16_256p_valid(p, ks)) return hipErrorInvalidValue;
  p.ntm = (p.M + 255) / 256;
  p.ntn = p.Cout / 256;
  const DeviceInfo &di = device_info();   // CU count of this device + the > 64 KB dynamic-LDS opt-in
  if (di.status != hipSuccess) return di.status;
  const int ntiles = p.ntm * p.ntn;
  const int slots = di.n_cu & ~7;          // a multiple of 8: a workgroup's tiles then all sit in its own XCD's chunk
  const dim3 grid((unsigned)(ntiles < slots || slots < 8 ? ntiles : slots)), block(512);
  if (ks == 3) hipLaunchKernelGGL((conv_bf16_256p_kernel<3, false>), grid, block, kLds256pBytes, s, p);
  else if (p.T > 0) hipLaunchKernelGGL((conv_bf16_256p_kernel<1, true>), grid, block, kLds256pBytes, s, p);
  else if (p.res) hipLaunchKernelGGL((conv_bf16_256p_kernel<1, false, true, false>), grid, block, kLds256pBytes, s, p);
  else if (p.x2) hipLaunchKernelGGL((conv_bf16_256p_kernel<1, false, false, true>), grid, block, kLds256pBytes, s, p);
  else hipLaunchKernelGGL((conv_bf16_256p_kernel<1, false>), grid, block, kLds256pBytes, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conv3x3_ws: the bf16 3x3 (stride 1, pad 1) convolution with 64 input and 64 output channels -- Bottleneck.conv2 of
// layer1 -- as a WEIGHT-STATIONARY direct convolution.
//
// Why: with Cout = 64 an LDS-staged implicit GEMM reads (TM + TN) fragments per TM x TN MFMAs with TN <= 2; the
// 128 x 64 tile of conv_igemm needs 1.5 ds_read_b128 per MFMA (one per MFMA already saturates the LDS port: four SIMDs x
// 1 KiB per 32-cycle MFMA = 128 B/clk) and re-stages every input pixel nine times: 0.20 of the bf16 MFMA peak in
// profiles/r02_bf16_config5_per_layer.txt, the slowest launches of the config-5 forward after the stem.  Here:
//   * ONE workgroup of four waves per CU (one wave per SIMD, the whole 512-register file each), persistent over tiles;
//     every wave keeps ALL of W2 -- 64 x 576 bf16 = 72 fragments = 288 registers -- for the life of the kernel, so the
//     only LDS reads are the pixel fragments: 0.5 ds_read_b128 per MFMA, no weight traffic at all after the prologue;
//   * the input of a tile of TR x TC output pixels (<= 256) is its (TR + 2) x (TC + 2) halo patch, brought in ONCE by
//     LDS-DMA (`buffer_load ... lds`, 8 pixels = 1 KiB per wave-instruction, zero padding by the descriptor's range
//     check / an out-of-range offset) into one of two buffers: the patch of tile i + 1 lands while tile i is computed;
//     all nine taps read it from LDS -- no im2col re-staging.  A pixel's eight 16-byte chunks are XOR-swizzled by
//     (patch index & 7) on the SOURCE side; the fragment reads apply the same involution (conflict-free ds_read_b128);
//   * the product is computed TRANSPOSED -- A = weights (rows = output channels), B = pixels (columns) -- so a lane ends
//     up with ONE pixel and 16 channels in groups of four: after bias / ReLU / bf16 two `v_permlane32_swap` per group
//     pair make whole 16-byte channel groups, stored straight from registers (no LDS round trip in the epilogue);
//   * one barrier per tile; the DMA of the next patch is retired (vmcnt) just before the last stores of the tile are
//     issued, so no wait ever sees a store it has just issued.
// Per output the products enter the fp32 accumulator in conv_igemm's order (taps ascending, k16 groups ascending, the
// same eight k per lane half; a*b commutes), so results are bit-identical to the other bf16 tiles.
// ---------------------------------------------------------------------------------------------
constexpr int kWsRounds = 11;                    // DMA rounds of 32 patch pixels (4 waves x 8 pixels)
constexpr int kWsPatchMax = kWsRounds * 32;      // 352 patch pixels per buffer (18 x 18 for a 16 x 16 tile, 6 x 58 for 4 x 56)
constexpr int kWsPlane = kWsPatchMax * 32;      // one k16 group of every patch pixel
constexpr int kWsBufBytes = 4 * kWsPlane;        // 45 056 B
constexpr int kWsLdsBytes = 2 * kWsBufBytes + 256;
constexpr int kWsTableOff = kWsBufBytes + 256 + 1024 + 32768;      // FUSE3: one patch buffer, bias2, bias3, conv3's weights in fragment order
constexpr int kWsRingOff = kWsTableOff + kWsRounds * 1024;         // ... the loader's per-thread offset table
constexpr int kWsSlots = 4;                                        // residual ring: slots of 4 KB (one 32-channel tile of the wave's 64 pixels) per wave
constexpr int kWsLaneOff = kWsRingOff + 4 * kWsSlots * 4096;       // ... six per-thread tile-invariant words (pixel positions)
constexpr int kWsLdsBytes3All = kWsLaneOff + 6 * 1024;             // 162 048 B
static_assert(kWsLdsBytes3All <= 160 * 1024, "LDS budget of the fused weight-stationary kernel");
constexpr int kWsAgprFrags1 = 20;                // fragments of the second output-channel tile kept in accumulation registers

// Tile geometry for an H x W frame: TR x TC <= 256 output pixels, (TR + 2) x (TC + 2) <= kWsPatchMax patch pixels,
// fewest tiles per frame (ties: the smaller patch).  Returns false when nothing fits.
static bool ws_tile_geometry(int H, int W, int *tr_out, int *tc_out, int max_px = 256, int max_patch = kWsPatchMax) {
  long best_tiles = -1;
  int best_tr = 0, best_tc = 0, best_patch = 0;
  for (int tc = 4; tc <= 128; ++tc) {
    int tr = max_px / tc;
    if (tr > H) tr = H;
    if (tr < 1) continue;
    const int patch = (tr + 2) * (tc + 2);
    if (patch > max_patch) continue;
    const long tiles = (long)((H + tr - 1) / tr) * ((W + tc - 1) / tc);
    if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && patch < best_patch)) {
      best_tiles = tiles; best_tr = tr; best_tc = tc; best_patch = patch;
    }
  }
  *tr_out = best_tr;
  *tc_out = best_tc;
  return best_tiles > 0;
}

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N - 1>) -- unrolled in the AST, for bodies too large for
// `#pragma unroll` to honour (its size threshold silently leaves a loop, and the register arrays go to scratch).
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// `s_waitcnt vmcnt(n)` for a compile-time-foldable n (the instruction takes an immediate).
__device__ __forceinline__ void wait_vmcnt(int n) {
#define TSM_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    TSM_VMCNT_CASE(0) TSM_VMCNT_CASE(4) TSM_VMCNT_CASE(8) TSM_VMCNT_CASE(12) TSM_VMCNT_CASE(16) TSM_VMCNT_CASE(20)
    TSM_VMCNT_CASE(24) TSM_VMCNT_CASE(28) TSM_VMCNT_CASE(32) TSM_VMCNT_CASE(36) TSM_VMCNT_CASE(40) TSM_VMCNT_CASE(44)
    TSM_VMCNT_CASE(48) TSM_VMCNT_CASE(52) TSM_VMCNT_CASE(56) TSM_VMCNT_CASE(60)
    TSM_VMCNT_CASE(3) TSM_VMCNT_CASE(7) TSM_VMCNT_CASE(11) TSM_VMCNT_CASE(15) TSM_VMCNT_CASE(19) TSM_VMCNT_CASE(23)
    TSM_VMCNT_CASE(27) TSM_VMCNT_CASE(31) TSM_VMCNT_CASE(35) TSM_VMCNT_CASE(39) TSM_VMCNT_CASE(43) TSM_VMCNT_CASE(47)
    TSM_VMCNT_CASE(51) TSM_VMCNT_CASE(55) TSM_VMCNT_CASE(59) TSM_VMCNT_CASE(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef TSM_VMCNT_CASE
}
// Vector-memory operations of the fused kernel's conv3 phase that are younger than the ring fill R_it when tile `it`
// waits for it.  Issue order with D = kWsSlots:  R_0 .. R_{D-1} | P (kWsRounds) | [it = 0: wait, R_D, S_0] [1: wait,
// R_{D+1}, S_1] ..., R = 4 fills of one slot (only while it + D < 8), S = 4 stores.
constexpr int ws_younger_than_fill(int it) {
  int n = 0;
  if (it < kWsSlots) n += (kWsSlots - 1 - it) * 4 + kWsRounds;         // the rest of the prologue fills, the patch
  else n += 4;                                                          // S_{it - D}, issued right after R_it
  for (int i = (it < kWsSlots ? 0 : it - kWsSlots + 1); i < it; ++i) n += (i + kWsSlots < 8 ? 4 : 0) + 4;
  return n;
}

// One parameter block for both forms of the kernel: FUSE3 = false, the 3x3 conv alone (y = [M][64]); FUSE3 = true,
// Bottleneck.conv2 + bn2 + ReLU + conv3 + bn3 + residual + ReLU (y, res = [M][256]).
struct WsParams {
  const void *x;       // [N, H, W, 64] bf16
  const void *w2;      // [64][576] bf16, K = (ky, kx, c), BN scale folded in
  const float *bias2;  // [64]
  const void *w3;      // FUSE3: [256][64] bf16 (conv3's packed weights, row-major)
  const float *bias3;  // FUSE3: [256]
  const void *res;     // FUSE3: [M, 256] bf16, the block input
  void *y;
  int N, H, W, M, relu, reverse, tr, tc;
};

// FUSE3: conv3 rides behind conv2 in the same registers.  With the transposed product a lane of conv2's accumulator
// holds ONE pixel and 4 consecutive mid channels per group; after bias / ReLU / bf16 one v_permlane32_swap per word
// pairs the two lane halves into 8 consecutive channels = exactly the B fragment (k16 group) of the next MFMA: the
// 64-channel mid tensor never leaves the register file (no LDS, no HBM).  conv3 is again transposed (A = W3 fragments,
// read from an LDS copy in fragment order: one read feeds the MFMAs of both M-tiles of the wave), its epilogue adds
// bias3 and the residual in the accumulator layout (the residual arrives as 16-byte groups and goes through the same
// swap backwards), and stores 16-byte groups.  Same products in the same order per accumulator as the two separate
// launches (conv3: k16 groups ascending over its 64 channels), same epilogue arithmetic: bit-identical to them.
template <bool FUSE3>
__global__ void __launch_bounds__(256, 1) conv3x3_ws_kernel(const WsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kWsBufBytes | bias2 | (FUSE3: bias3 | W3 fragments)
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W, TR = p.tr, TC = p.tc, PW = TC + 2;
  const int nr = ((TR + 2) * PW + 31) >> 5;                             // DMA rounds in use (<= kWsRounds)
  const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR, tiles_f = tiles_x * tiles_y;
  const int ntiles = p.N * tiles_f;
  const int frame_bytes = H * W * 128;

  // ---- the stationary operand: fragment s = tap * 4 + g of output-channel tile nt, k = 16 s + 8 half .. + 8.
  // 56 of the 72 fragments are pinned to the accumulation-register half of the file (MFMA reads them there directly).
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, 64 * 576 * 2, 0x00020000);
  u32x4 wr[2][36];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int s = 0; s < 36; ++s)
      wr[nt][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, ((nt * 32 + l31) * 576 + s * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int s = 0; s < 36; ++s)
      if (nt == 0 || s < kWsAgprFrags1) asm volatile("" : "+a"(wr[nt][s]));
  // LDS: !FUSE3  two patch buffers | bias2;   FUSE3  one patch buffer | bias2 | bias3 | W3 fragments | loader table | residual rings
  float *bias_lds = reinterpret_cast<float *>(lds + (FUSE3 ? 1 : 2) * kWsBufBytes);
  float *bias3_lds = bias_lds + 64;
  unsigned char *w3_lds = lds + kWsBufBytes + 256 + 1024;               // [it * 4 + g][lane] 16 B: conv3's A fragments
  if (tid < 64) bias_lds[tid] = p.bias2[tid];
  if constexpr (FUSE3) {
    bias3_lds[tid] = p.bias3[tid];
    const __amdgpu_buffer_rsrc_t rsrcW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w3), 0, 256 * 64 * 2, 0x00020000);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int f = wave * 8 + k, it = f >> 2, g = f & 3;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrcW3, ((32 * it + l31) * 64 + 16 * g + 8 * half) * 2, 0, 0);
      *reinterpret_cast<u32x4 *>(w3_lds + f * 1024 + lane * 16) = v;
    }
  }
  const float floor_ = (FUSE3 || p.relu) ? 0.f : -INFINITY;

  // ---- loader state.  A buffer is four planes, one per k16 group g: plane g holds bytes [32 g, 32 g + 32) of every patch
  // pixel, 32 B per pixel, the two 16-byte halves swapped where (pixel >> 3) is odd.  Wave w fills plane w: in round i
  // its lane fills half (lane & 1) of patch pixel 32 i + (lane >> 1).
  const int chunk = 2 * wave + ((lane & 1) ^ ((lane >> 4) & 1));        // source chunk of that half
  // per round: (byte offset of the chunk relative to the patch origin) >> 4 | patch column << 24 -- in registers, or
  // (FUSE3, whose conv3 phase needs them for other things) in a per-thread LDS table
  unsigned dslot[FUSE3 ? 1 : kWsRounds];
  unsigned *dslot_lds = reinterpret_cast<unsigned *>(lds + kWsTableOff) + tid;
#pragma unroll
  for (int i = 0; i < kWsRounds; ++i) {
    const int pidx = 32 * i + (lane >> 1);
    const int pr = pidx / PW, pc = pidx - pr * PW;
    const unsigned v = (unsigned)((pr * W + pc) * 8 + chunk) | ((unsigned)pc << 24);
    if constexpr (FUSE3) dslot_lds[i * 256] = v;
    else dslot[i] = v;
  }
  auto issue_patch = [&](int t, int b) {
    const int f = t / tiles_f, rem = t - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x0 = tx * TC;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)f * frame_bytes), 0, frame_bytes, 0x00020000);
    const int tbase = ((ty * TR - 1) * W + (x0 - 1)) * 128;             // rows above / below the frame fall outside the descriptor: zeros
    unsigned char *dst = lds + b * kWsBufBytes + wave * kWsPlane;
#pragma unroll
    for (int i = 0; i < kWsRounds; ++i)
      if (i < nr) {
        const unsigned ds = FUSE3 ? dslot_lds[i * 256] : dslot[FUSE3 ? 0 : i];
        const int xg = x0 - 1 + (int)(ds >> 24);
        const unsigned off = (unsigned)tbase + ((ds & 0xFFFFFFu) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(dst + i * 1024), 16,
                                                 (int)((unsigned)xg < (unsigned)W ? off : kInvalid), 0, 0, 0);
      }
  };

  // FUSE3: the same with a CONSTANT number of operations (rounds past the patch, or t < 0 = no next tile, fetch nothing:
  // an out-of-range offset writes zeros) -- its counted waits depend on it; always into the one buffer.
  auto issue_patch_full = [&](int t) {
    const int tq = t < 0 ? 0 : t;
    const int f = tq / tiles_f, rem = tq - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x0 = tx * TC;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)f * frame_bytes), 0, frame_bytes, 0x00020000);
    const int tbase = ((ty * TR - 1) * W + (x0 - 1)) * 128;
    unsigned char *dst = lds + wave * kWsPlane;
#pragma unroll
    for (int i = 0; i < kWsRounds; ++i) {
      const unsigned ds = dslot_lds[i * 256];
      const int xg = x0 - 1 + (int)(ds >> 24);
      const unsigned off = (unsigned)tbase + ((ds & 0xFFFFFFu) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(dst + i * 1024), 16,
                                               (int)((t >= 0 && i < nr && (unsigned)xg < (unsigned)W) ? off : kInvalid), 0, 0, 0);
    }
  };

  // ---- this lane's two output pixels (M-tile mt = 0, 1 of the wave): position in the tile and in the patch
  int prow[2], pcol[2], pp0[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int q = wave * 64 + mt * 32 + l31;
    const bool ok = q < TR * TC;
    const int r = q / TC, c = q - r * TC;
    prow[mt] = ok ? r : 0x4000;                                         // (a row no frame has: the store is dropped)
    pcol[mt] = c;
    pp0[mt] = ok ? r * PW + c : 0;
  }
  // !FUSE3: one descriptor over the whole output (stores of a tile are issued while the next one is computed)
  const __amdgpu_buffer_rsrc_t rsrcYall = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * 128), 0x00020000);

  // The accumulators of one 32-pixel M-tile -> bf16, in ten pieces (they ride on the MFMA steps of the NEXT M-tile):
  // lane = pixel, a[nt][4 q + j] = channel nt * 32 + 8 q + 4 half + j.  Pieces 0-3 / 5-8: bias, ReLU, bf16 of group q of
  // nt = 0 / 1.  Pieces 4 / 9, !FUSE3: lanes 0-31 take groups 0, 1 and lanes 32-63 groups 2, 3 of the pixel (one
  // v_permlane32_swap per word brings the other half's words in) and store them as whole 16-byte groups.  FUSE3: the
  // swap pairs groups (0, 1) and (2, 3) instead: lanes 0-31 then hold channels 16 g' .. + 8 and lanes 32-63 channels
  // 16 g' + 8 .. + 8 of k16 group g' = 2 nt, 2 nt + 1 -- conv3's B fragments, kept in `mid`.
  unsigned pk[4][2];
  auto epi_piece = [&](const f32x16 (&a)[2], int k, unsigned yoff, u32x4 *mid) {
    const int nt = k / 5, q = k - nt * 5;
    if (q < 4) {
      const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + nt * 32 + 8 * q + 4 * half);
      pk[q][0] = pack_bf16(fmaxf(a[nt][4 * q] + b[0], floor_), fmaxf(a[nt][4 * q + 1] + b[1], floor_));
      pk[q][1] = pack_bf16(fmaxf(a[nt][4 * q + 2] + b[2], floor_), fmaxf(a[nt][4 * q + 3] + b[3], floor_));
    } else if constexpr (FUSE3) {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][0], pk[2 * qq + 1][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][1], pk[2 * qq + 1][1], false, false);
        mid[2 * nt + qq] = u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
    } else {
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
        const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcYall, (int)(yoff + (unsigned)(nt * 64 + (2 * half + qq) * 16)), 0, 0);
      }
    }
  };
  auto out_off = [&](int tt, int mt) -> unsigned {                      // !FUSE3: byte offset of this lane's pixel of tile tt, or dropped
    const int f = tt / tiles_f, rem = tt - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy = ty * TR + prow[mt], ox = tx * TC + pcol[mt];
    return (oy < H && ox < W) ? (unsigned)(((f * H + oy) * W + ox) * 128) : kInvalid;
  };

  // One M-tile: 36 steps (tap, g) of one pixel-fragment read (three steps ahead) and two MFMAs; the pieces of the
  // PREVIOUS M-tile (accumulators `prev`), if any, are spread over steps 2, 5, .., 29.
  auto mtile = [&](const unsigned char *buf, int mt, f32x16 (&acc)[2], const f32x16 (&prev)[2], bool has_prev, unsigned prev_off,
                   u32x4 *prev_mid) {
    u32x4 px[4];
    unsigned tb = 0;
    auto rd = [&](int s) {
      const int tap = s >> 2, g = s & 3, ky = tap / 3, kx = tap - ky * 3;
      if (g == 0) {
        const int pp = pp0[mt] + ky * PW + kx;
        tb = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));
      }
      px[s & 3] = *reinterpret_cast<const u32x4 *>(buf + tb + g * kWsPlane);
    };
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    rd(0); rd(1); rd(2);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      if (s + 3 < 36) rd(s + 3);
      if (has_prev && s >= 2 && s < 32 && (s - 2) % 3 == 0) epi_piece(prev, (s - 2) / 3, prev_off, prev_mid);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[nt][s]), __builtin_bit_cast(bf16x8, px[s & 3]),
                                                          acc[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x16 accA[2], accB[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[nt][e] = 0.f;
  int t = blockIdx.x, nb = 0;
  if (t < ntiles) issue_patch(p.reverse ? ntiles - 1 - t : t, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           // the weights, this wave's share of the first patch, its LDS fills

  if constexpr (!FUSE3) {
    unsigned offB = kInvalid;                                           // nothing to store before the first tile
    for (; t < ntiles; t += gridDim.x, nb ^= 1) {
      __builtin_amdgcn_s_barrier();    // every wave's share of this patch has landed; nobody still reads the other buffer
      const int tn = t + gridDim.x;
      if (tn < ntiles) issue_patch(p.reverse ? ntiles - 1 - tn : tn, nb ^ 1);
      const int tt = p.reverse ? ntiles - 1 - t : t;
      const unsigned char *buf = lds + nb * kWsBufBytes;
      mtile(buf, 0, accA, accB, true, offB, nullptr);                   // (B = M-tile 1 of the previous tile)
      const unsigned offA = out_off(tt, 0);
      mtile(buf, 1, accB, accA, true, offA, nullptr);
      offB = out_off(tt, 1);
      // the next patch is older than the eight stores this iteration issued: retire it, not them
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) epi_piece(accB, k, offB, nullptr);     // M-tile 1 of the last tile
  } else {
    // FUSE3 uses ONE patch buffer (the conv3 phase does not read it: the next patch is fetched under that phase) and
    // spends the LDS on a wave-private ring of kWsSlots residual slots, filled by LDS-DMA that many output-channel
    // tiles ahead of their use: 16 KB of residual in flight per wave without a register.  A slot holds the wave's 64 pixels x
    // 64 B (one tile of 32 channels); chunk c of pixel x sits at 16-byte position (c + (x >> 2)) & 3 of its row (swizzle
    // on the source side; the ds_read_b64 of the accumulator layout -- lane = pixel, 4 channels -- is conflict-free).
    const int frame_out = H * W * 512;
    unsigned char *ring = lds + kWsRingOff + wave * (kWsSlots * 4096);
    // loader lanes of the ring: lane fills position (lane & 3) of pixel 16 j + (lane >> 2), j = 0..3
    const int rchunk = ((lane & 3) - (lane >> 4)) & 3;
    // per-thread words kept in LDS (the registers are spent on weights): [0..3] the loader pixel's tile row | column << 16
    // (row 0x4000: not in the tile), [4..5] the same for this lane's pixel of M-tile 0 / 1
    int *lane_lds = reinterpret_cast<int *>(lds + kWsLaneOff) + tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wave * 64 + 16 * j + (lane >> 2);
      const int r = q / TC, c = q - r * TC;
      lane_lds[j * 256] = (q < TR * TC ? r : 0x4000) | (c << 16);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) lane_lds[(4 + mt) * 256] = prow[mt] | (pcol[mt] << 16);
    unsigned rrd[2];                                                    // read offset of this lane's pixel of M-tile mt, group q = 0 (+ 16 ((q + s) & 3) - 16 s per q)
    const int rsw = (l31 >> 2) & 3;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) rrd[mt] = (unsigned)((mt * 32 + l31) * 64 + half * 8);
    for (; t < ntiles; t += gridDim.x) {
      const int tt = p.reverse ? ntiles - 1 - t : t;
      const int f = tt / tiles_f, rem = tt - f * tiles_f;
      const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
      const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.res) + (size_t)f * frame_out), 0, frame_out, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char *>(p.y) + (size_t)f * frame_out, 0, frame_out, 0x00020000);
      unsigned yo[2], ro[4];       // byte offset of the 256 channels of: this lane's pixel of M-tile mt / its loader pixel j (+ its chunk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rc = lane_lds[(4 + mt) * 256];
        const int oy = ty * TR + (rc & 0xFFFF), ox = tx * TC + (rc >> 16);
        yo[mt] = (oy < H && ox < W) ? (unsigned)((oy * W + ox) * 512) : kInvalid;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rc = lane_lds[j * 256];
        const int oy = ty * TR + (rc & 0xFFFF), ox = tx * TC + (rc >> 16);
        ro[j] = (oy < H && ox < W) ? (unsigned)((oy * W + ox) * 512 + rchunk * 16) : kInvalid;
      }
      auto issue_res = [&](int it) {                                    // 4 vector-memory operations
        unsigned char *dst = ring + (it % kWsSlots) * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcR, (lds_void *)(dst + j * 1024), 16,
                                                   (int)(ro[j] == kInvalid ? kInvalid : ro[j] + (unsigned)(it * 64)), 0, 0, 0);
      };
      // (the eight youngest operations are stores of the previous tile; the patch is older)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();    // every wave's share of this patch has landed
      #pragma unroll
      for (int i = 0; i < kWsSlots; ++i) issue_res(i);
      u32x4 mid[2][4];
      mtile(lds, 0, accA, accB, false, 0u, nullptr);
      mtile(lds, 1, accB, accA, true, 0u, mid[0]);
#pragma unroll
      for (int k = 0; k < 10; ++k) epi_piece(accB, k, 0u, mid[1]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();    // nobody reads the patch any more: fetch the next one under the conv3 phase
      {                                // always kWsRounds operations (dead ones past the patch / past the last tile), so that the waits below count
        const int tn = t + gridDim.x;
        issue_patch_full(tn < ntiles ? (p.reverse ? ntiles - 1 - tn : tn) : -1);
      }
      // ---- conv3: eight tiles of 32 output channels, both M-tiles per W3 fragment; the wait of tile `it` leaves exactly
      // the vector-memory operations younger than its ring fill in flight (ws_younger_than_fill)
      // (software-pipelined: the eight MFMAs of tile it + 1 are issued before the epilogue of tile it and run under it)
      f32x16 c3[2][2];
      auto conv3_mfma = [&](int it, f32x16 (&c)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e) c[mt][e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(w3_lds + (it * 4 + g) * 1024 + lane * 16));
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            c[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(bf16x8, mid[mt][g]), c[mt], 0, 0, 0);
        }
      };
      conv3_mfma(0, c3[0]);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        wait_vmcnt(ws_younger_than_fill(it));
        // the residual in the accumulator layout: rp[mt][q] = channels it * 32 + 8 q + 4 half .. + 4 of this lane's pixel
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 rp[2][4];
        const unsigned char *slot = ring + (it % kWsSlots) * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            rp[mt][q] = *reinterpret_cast<const u32x2 *>(slot + rrd[mt] + (((q + rsw) & 3) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the slot is free again
        if (it + kWsSlots < 8) issue_res(it + kWsSlots);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < 8) conv3_mfma(it + 1, c3[(it + 1) & 1]);
        f32x16 (&cc)[2] = c3[it & 1];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          // + bias3, + residual, ReLU, bf16; lanes 0-31 then take groups 0, 1 and lanes 32-63 groups 2, 3 of the pixel
          // (v_permlane32_swap) and store them as 16-byte groups, straight from registers
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias3_lds + it * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {                // two channels at a time: packed fp32 adds (v_pk_add_f32)
              const unsigned rw = rp[mt][q][w2];
              f32x2 v = f32x2{cc[mt][4 * q + 2 * w2], cc[mt][4 * q + 2 * w2 + 1]} + f32x2{b[2 * w2], b[2 * w2 + 1]};
              v += f32x2{__builtin_bit_cast(float, rw << 16), __builtin_bit_cast(float, rw & 0xFFFF0000u)};
              pk[q][w2] = pack_bf16(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
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
            const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
            __builtin_amdgcn_raw_buffer_store_b128(
                o, rsrcY, (int)(yo[mt] == kInvalid ? kInvalid : yo[mt] + (unsigned)(it * 64 + (2 * half + qq) * 16)), 0, 0);
          }
        }
        // schedule of this region: one W3 fragment read, then its two MFMAs, each followed by a share of the epilogue's
        // vector ALU work (in program order the eight MFMAs would be issued back to back and stall the wave on the pipe)
        if (it + 1 < 8) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  // VALU
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// conv3x3_ws128: the weight-stationary form for 128 -> 128 channels (Bottleneck.conv2 of layer2, bf16).  W2 is
// 128 x 1152 bf16 = 288 KB: no wave can hold it, so the OUTPUT CHANNELS are split over the four waves -- wave w keeps the
// 72 fragments (288 registers) of channels 32 w .. 32 w + 31 for all of K -- and every wave walks ALL pixels of the
// tile: 128 pixels = 4 M-tiles, two at a time (two independent accumulator chains), one pixel-fragment read per MFMA
// (K is never split across waves: every output still accumulates its 1152 products in conv_igemm's order, bit-identical).
// The rest is conv3x3_ws_kernel<false>: persistent workgroups, the (TR + 2) x (TC + 2) patch of 256-byte pixels by
// LDS-DMA into one of two buffers of eight 32-byte planes (wave w fills planes 2 w, 2 w + 1), transposed MFMA, the
// epilogue of a pair of M-tiles in ten pieces under the MFMA steps of the next pair, 16-byte groups stored from
// registers (a wave writes its own 64-byte channel slice of each pixel).
// ---------------------------------------------------------------------------------------------
constexpr int kW8Rounds = 6;                      // DMA rounds of 32 patch pixels per plane
constexpr int kW8PatchMax = kW8Rounds * 32;       // 192 patch pixels (10 x 18 for an 8 x 16 tile, 6 x 30 for 4 x 28)
constexpr int kW8Plane = kW8PatchMax * 32;
constexpr int kW8BufBytes = 8 * kW8Plane;         // 49 152 B
constexpr int kW8LdsBytes = 2 * kW8BufBytes + 512;
constexpr int kW8AgprFrags = 56;                  // fragments kept in accumulation registers

__global__ void __launch_bounds__(256, 1) conv3x3_ws128_kernel(const WsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kW8BufBytes | bias
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W, TR = p.tr, TC = p.tc, PW = TC + 2;
  const int nr = ((TR + 2) * PW + 31) >> 5;
  const int tiles_x = (W + TC - 1) / TC, tiles_y = (H + TR - 1) / TR, tiles_f = tiles_x * tiles_y;
  const int ntiles = p.N * tiles_f;
  const int frame_bytes = H * W * 256;

  // ---- the stationary operand: fragment s = tap * 8 + g of this wave's 32 output channels, k = 16 s + 8 half .. + 8
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, 128 * 1152 * 2, 0x00020000);
  u32x4 wr[72];
#pragma unroll
  for (int s = 0; s < 72; ++s)
    wr[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, ((wave * 32 + l31) * 1152 + s * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int s = 0; s < kW8AgprFrags; ++s) asm volatile("" : "+a"(wr[s]));
  float *bias_lds = reinterpret_cast<float *>(lds + 2 * kW8BufBytes);
  if (tid < 128) bias_lds[tid] = p.bias2[tid];
  const float floor_ = p.relu ? 0.f : -INFINITY;

  // ---- loader: plane g holds bytes [32 g, 32 g + 32) of every patch pixel (halves swapped where (pixel >> 3) is odd);
  // in round i this lane fills half (lane & 1) of patch pixel 32 i + (lane >> 1), in planes 2 wave and 2 wave + 1
  const int hsel = (lane & 1) ^ ((lane >> 4) & 1);
  unsigned dslot[kW8Rounds];                       // (byte offset of the pixel relative to the patch origin) >> 4 | patch column << 24
#pragma unroll
  for (int i = 0; i < kW8Rounds; ++i) {
    const int pidx = 32 * i + (lane >> 1);
    const int pr = pidx / PW, pc = pidx - pr * PW;
    dslot[i] = (unsigned)((pr * W + pc) * 16) | ((unsigned)pc << 24);
  }
  auto issue_patch = [&](int t, int b) {
    const int f = t / tiles_f, rem = t - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x0 = tx * TC;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)f * frame_bytes), 0, frame_bytes, 0x00020000);
    const int tbase = ((ty * TR - 1) * W + (x0 - 1)) * 256 + (4 * wave + hsel) * 16;
    unsigned char *dst = lds + b * kW8BufBytes + 2 * wave * kW8Plane;
#pragma unroll
    for (int i = 0; i < kW8Rounds; ++i)
      if (i < nr) {
        const int xg = x0 - 1 + (int)(dslot[i] >> 24);
        const unsigned off = (unsigned)tbase + ((dslot[i] & 0xFFFFFFu) << 4);
        const unsigned o = (unsigned)xg < (unsigned)W ? off : kInvalid;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(dst + i * 1024), 16, (int)o, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(dst + kW8Plane + i * 1024), 16,
                                                 (int)(o == kInvalid ? kInvalid : o + 32u), 0, 0, 0);
      }
  };

  // ---- this lane's pixel in each of the four M-tiles (shared by all waves)
  int prow[4], pcol[4], pp0[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int q = mt * 32 + l31;
    const bool ok = q < TR * TC;
    const int r = q / TC, c = q - r * TC;
    prow[mt] = ok ? r : 0x4000;
    pcol[mt] = c;
    pp0[mt] = ok ? r * PW + c : 0;
  }
  const __amdgpu_buffer_rsrc_t rsrcYall = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * 256), 0x00020000);

  // epilogue of a PAIR of M-tiles in ten pieces: per M-tile four (bias, ReLU, bf16 of group q) and one (lanes 0-31 take
  // groups 0, 1, lanes 32-63 groups 2, 3: swap, two 16-byte stores into this wave's 64-byte slice of the pixel)
  unsigned pk[4][2];
  auto epi_piece = [&](const f32x16 (&a)[2], int k, const unsigned (&yoff)[2]) {
    const int m = k / 5, q = k - m * 5;
    if (q < 4) {
      const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + wave * 32 + 8 * q + 4 * half);
      pk[q][0] = pack_bf16(fmaxf(a[m][4 * q] + b[0], floor_), fmaxf(a[m][4 * q + 1] + b[1], floor_));
      pk[q][1] = pack_bf16(fmaxf(a[m][4 * q + 2] + b[2], floor_), fmaxf(a[m][4 * q + 3] + b[3], floor_));
    } else {
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
        const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
        __builtin_amdgcn_raw_buffer_store_b128(
            o, rsrcYall, (int)(yoff[m] == kInvalid ? kInvalid : yoff[m] + (unsigned)(wave * 64 + (2 * half + qq) * 16)), 0, 0);
      }
    }
  };
  auto out_off = [&](int tt, int mt) -> unsigned {
    const int f = tt / tiles_f, rem = tt - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy = ty * TR + prow[mt], ox = tx * TC + pcol[mt];
    return (oy < H && ox < W) ? (unsigned)(((f * H + oy) * W + ox) * 256) : kInvalid;
  };

  // A pair of M-tiles (2 mp, 2 mp + 1): 72 steps (tap, g) of two pixel-fragment reads (two steps ahead) and two MFMAs
  // with the same weight fragment; the pieces of the previous pair are spread over steps 3, 10, .., 66.
  auto mpair = [&](const unsigned char *buf, int mp, f32x16 (&acc)[2], const f32x16 (&prev)[2], const unsigned (&prev_off)[2]) {
    u32x4 px[4][2];
    unsigned tb[2] = {0u, 0u};
    auto rd = [&](int s) {
      const int tap = s >> 3, g = s & 7, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (g == 0) {
          const int pp = pp0[2 * mp + m] + ky * PW + kx;
          tb[m] = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));
        }
        px[s & 3][m] = *reinterpret_cast<const u32x4 *>(buf + tb[m] + g * kW8Plane);
      }
    };
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    rd(0); rd(1); rd(2);
    static_for<72>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s + 3 < 72) rd(s + 3);
      if constexpr (s >= 3 && s < 70 && (s - 3) % 7 == 0) epi_piece(prev, (s - 3) / 7, prev_off);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[s]), __builtin_bit_cast(bf16x8, px[s & 3][m]),
                                                         acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  f32x16 accA[2], accB[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[m][e] = 0.f;
  unsigned offA[2], offB[2] = {kInvalid, kInvalid};                     // nothing to store before the first tile
  int t = blockIdx.x, nb = 0;
  if (t < ntiles) issue_patch(p.reverse ? ntiles - 1 - t : t, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (; t < ntiles; t += gridDim.x, nb ^= 1) {
    __builtin_amdgcn_s_barrier();      // every wave's share of this patch has landed; nobody still reads the other buffer
    const int tn = t + gridDim.x;
    if (tn < ntiles) issue_patch(p.reverse ? ntiles - 1 - tn : tn, nb ^ 1);
    const int tt = p.reverse ? ntiles - 1 - t : t;
    const unsigned char *buf = lds + nb * kW8BufBytes;
    mpair(buf, 0, accA, accB, offB);                                    // (B = M-tiles 2, 3 of the previous tile)
    offA[0] = out_off(tt, 0); offA[1] = out_off(tt, 1);
    mpair(buf, 1, accB, accA, offA);
    offB[0] = out_off(tt, 2); offB[1] = out_off(tt, 3);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                    // the next patch is older than this iteration's eight stores
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) epi_piece(accB, k, offB);
}

bool conv3x3_ws128_valid(const ConvParams &p) {
  int tr, tc;
  return p.prec == kPrecBf16 && p.C == 128 && p.Cout == 128 && p.Kp == 1152 && p.stride == 1 && p.pad == 1 && p.Hi == p.Ho &&
         p.Wi == p.Wo && !p.res && !p.x2 && p.T == 0 && p.kseg_len == 0 && (double)p.M * 256.0 < 2.0e9 &&
         ws_tile_geometry(p.Hi, p.Wi, &tr, &tc, 128, kW8PatchMax);
}

// ---------------------------------------------------------------------------------------------
// conv1x1_ws: Bottleneck.conv1 of layer1 in bf16 (1x1, CIN = 64 or 256 -> 64 channels, the temporal shift fused into
// the loader).  These launches are pure HBM streams (2.7 GB in 0.58 ms with conv_igemm's register-staged K loop, which
// waits for every K-step's loads: 4.6 TB/s); this form keeps W1 (32 / 8 fragments per wave) in registers and brings
// a tile's 128 pixels x CIN channels in by LDS-DMA, one whole tile (64 KB) ahead of its use, so the memory system
// always has a CU's next 64 KB in flight and nothing in the compute loop waits on it.  A tile = 128 consecutive rows of
// the flattened N*H*W; LDS = two buffers of CIN / 16 planes (plane g = bytes [32 g, 32 g + 32) of every pixel, halves
// swapped where (pixel >> 3) is odd).  The shift is an address choice per 16-byte chunk: channels < fold come from frame
// t + 1, < 2 fold from t - 1 (an out-of-range offset = zeros at the clip's ends).  Wave w multiplies pixels 32 w .. + 31
// (transposed MFMA: one pixel-fragment read feeds both output-channel tiles), k16 groups ascending = conv_igemm's
// order: bit-identical.  Epilogue from registers as in conv3x3_ws_kernel.
// ---------------------------------------------------------------------------------------------
struct Ws1Params {
  const void *x;       // [M, CIN] bf16
  const void *w;       // [64][CIN] bf16
  const float *bias;   // [64]
  void *y;             // [M, 64] bf16
  int M, HW, T, fold, relu, reverse;
};

template <int CIN>
__global__ void __launch_bounds__(256, 1) conv1x1_ws_kernel(const Ws1Params p) {
  constexpr int NG = CIN / 16;           // k16 groups = LDS planes
  constexpr int PPW = NG / 4;            // planes filled per wave
  constexpr int kPlane = 128 * 32;       // 128 pixels x 32 B
  constexpr int kBuf = NG * kPlane;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kBuf | bias
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntiles = (p.M + 127) >> 7;

  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, 64 * CIN * 2, 0x00020000);
  u32x4 wr[2][NG];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      wr[nt][g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, ((nt * 32 + l31) * CIN + g * 16 + half * 8) * 2, 0, 0);
  float *bias_lds = reinterpret_cast<float *>(lds + 2 * kBuf);
  if (tid < 64) bias_lds[tid] = p.bias[tid];
  const float floor_ = p.relu ? 0.f : -INFINITY;

  // loader: in round i (0..3) this lane fills half (lane & 1) of tile pixel 32 i + (lane >> 1), in planes PPW wave .. + PPW - 1
  const int hsel = (lane & 1) ^ ((lane >> 4) & 1);
  const int frame_bytes = p.HW * CIN * 2;
  auto issue_tile = [&](int t, int b) {
    const int m0 = t * 128;
    // descriptor rebased one frame before the tile: every offset below is small and non-negative
    const long base_row = (long)m0 - p.HW;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + base_row * (long)(CIN * 2)), 0,
        (int)((size_t)(128 + 2 * p.HW) * CIN * 2 > 0x7FFFFFF0u ? 0x7FFFFFF0u : (size_t)(128 + 2 * p.HW) * CIN * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pl = 32 * i + (lane >> 1);
      const int m = m0 + pl;
      const bool ok = m < p.M;
      const int n = (ok ? m : m0) / p.HW;
      const int tt = p.T > 0 ? n % p.T : 0;
      const unsigned own = (unsigned)((pl + p.HW) * CIN * 2);           // this pixel's row, relative to the rebased origin
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int g = PPW * wave + k;
        const int c0 = (2 * g + hsel) * 8;                              // first channel of this lane's 16-byte chunk
        unsigned off = own;
        bool valid = ok;
        if (p.T > 0 && c0 < p.fold) { off = own + (unsigned)frame_bytes; valid = ok && tt < p.T - 1; }
        else if (p.T > 0 && c0 < 2 * p.fold) { off = own - (unsigned)frame_bytes; valid = ok && tt > 0; }
        // (the first tile's "frame before" lies before the tensor: only ever addressed with valid == false)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(lds + b * kBuf + g * kPlane + i * 1024), 16,
                                                 (int)(valid ? off + (unsigned)(c0 * 2) : kInvalid), 0, 0, 0);
      }
    }
  };

  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.M * 128), 0x00020000);
  const int pp = wave * 32 + l31;                                        // this lane's pixel of the tile
  const unsigned rd = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));

  int t = blockIdx.x, nb = 0;
  if (t < ntiles) issue_tile(p.reverse ? ntiles - 1 - t : t, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (; t < ntiles; t += gridDim.x, nb ^= 1) {
    __builtin_amdgcn_s_barrier();      // this tile has landed (every wave waited for its share); the other buffer is free
    const int tn = t + gridDim.x;
    if (tn < ntiles) issue_tile(p.reverse ? ntiles - 1 - tn : tn, nb ^ 1);
    const int tt = p.reverse ? ntiles - 1 - t : t;
    const unsigned char *buf = lds + nb * kBuf;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const u32x4 px = *reinterpret_cast<const u32x4 *>(buf + rd + g * kPlane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[nt][g]), __builtin_bit_cast(bf16x8, px),
                                                          acc[nt], 0, 0, 0);
    }
    // epilogue: lane = pixel, acc[nt][4 q + j] = channel nt * 32 + 8 q + 4 half + j; lanes 0-31 store groups 0, 1 and
    // lanes 32-63 groups 2, 3 of the pixel (v_permlane32_swap), 16 bytes each
    const int m = tt * 128 + pp;
    const unsigned yoff = m < p.M ? (unsigned)m * 128u : kInvalid;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      unsigned pk[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + nt * 32 + 8 * q + 4 * half);
        pk[q][0] = pack_bf16(fmaxf(acc[nt][4 * q] + b[0], floor_), fmaxf(acc[nt][4 * q + 1] + b[1], floor_));
        pk[q][1] = pack_bf16(fmaxf(acc[nt][4 * q + 2] + b[2], floor_), fmaxf(acc[nt][4 * q + 3] + b[3], floor_));
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
        const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)(yoff == kInvalid ? kInvalid : yoff + (unsigned)(nt * 64 + (2 * half + qq) * 16)),
                                               0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the next tile is older than this tile's four stores
  }
}

bool conv1x1_ws_valid(const ConvParams &p) {
  return p.prec == kPrecBf16 && (p.C == 64 || p.C == 256) && p.Cout == 64 && p.Kp == p.C && p.stride == 1 && p.pad == 0 &&
         p.Hi == p.Ho && p.Wi == p.Wo && !p.res && !p.x2 && p.kseg_len == 0 && (double)p.M * 128.0 < 2.0e9 &&
         (p.T == 0 || (p.N % p.T == 0 && p.fold % 8 == 0 && 2 * p.fold <= p.C)) &&
         (double)(128 + 2.0 * p.Hi * p.Wi) * p.C * 2.0 < 2.0e9;
}

// conv1x1_wsn: the same streaming form for 128 / 256 output channels (conv1 of layer2 and of layer3.0, conv3 + downsample of
// layer1.0 as one GEMM over [conv3 input | block input]): the OUTPUT CHANNELS are split over the four waves (wave w keeps
// the fragments of channels COUT / 4 * w ..) and every wave multiplies all PX pixels of the tile.  PX = 128 (CIN <= 256)
// or 64 (CIN = 512): a tile buffer is 64 KB either way.  DUAL: chunks past K1 come from the second source (its own
// stride and frame size); SHIFT sources as in conv1x1_ws.
struct WsnParams {
  const void *x, *x2, *w;
  const float *bias;
  void *y;
  int M, HW, Wo, T, fold, relu, reverse;
  int K1;                    // channels of the first source (= CIN unless DUAL)
  int Hi2, Wi2, stride2;     // DUAL: second source [N, Hi2, Wi2, CIN - K1]
};

template <int CIN, int COUT, bool DUAL>
__global__ void __launch_bounds__(256, 1) conv1x1_wsn_kernel(const WsnParams p) {
  constexpr int PX = CIN <= 256 ? 128 : 64;
  constexpr int MT = PX / 32;
  constexpr int NG = CIN / 16;
  constexpr int NTW = COUT / 128;        // output-channel tiles per wave
  constexpr int PPW = NG / 4;            // planes filled per wave
  constexpr int kPlane = PX * 32;
  constexpr int kBuf = NG * kPlane;      // 64 KB
  constexpr int kAgpr = NTW * NG > 48 ? NTW * NG - 24 : 0;   // fragments pinned to accumulation registers (the 256-register form)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kBuf | bias
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntiles = (p.M + PX - 1) / PX;

  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, COUT * CIN * 2, 0x00020000);
  u32x4 wr[NTW][NG];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      wr[nt][g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (((wave * NTW + nt) * 32 + l31) * CIN + g * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (nt * NG + g < kAgpr) asm volatile("" : "+a"(wr[nt][g]));
  float *bias_lds = reinterpret_cast<float *>(lds + 2 * kBuf);
  if (tid < COUT) bias_lds[tid] = p.bias[tid];
  const float floor_ = p.relu ? 0.f : -INFINITY;

  const int hsel = (lane & 1) ^ ((lane >> 4) & 1);
  const int C1 = DUAL ? p.K1 : CIN, C2 = CIN - C1;
  const int frame_bytes = p.HW * C1 * 2;
  const int frame2_bytes = DUAL ? p.Hi2 * p.Wi2 * C2 * 2 : 0;
  auto issue_tile = [&](int t, int b) {
    const int m0 = t * PX;
    const long base_row = (long)m0 - p.HW;
    const size_t span = (size_t)(PX + 2 * p.HW) * C1 * 2;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + base_row * (long)(C1 * 2)), 0,
        (int)(span > 0x7FFFFFF0u ? 0x7FFFFFF0u : span), 0x00020000);
    const int n0 = m0 / p.HW;                                           // first frame of the tile
    const __amdgpu_buffer_rsrc_t rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(DUAL ? p.x2 : p.x) + (size_t)n0 * frame2_bytes), 0,
        DUAL ? (int)((size_t)(PX / p.HW + 2) * frame2_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : (size_t)(PX / p.HW + 2) * frame2_bytes) : 0,
        0x00020000);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int pl = 32 * i + (lane >> 1);
      const int m = m0 + pl;
      const bool ok = m < p.M;
      const int n = (ok ? m : m0) / p.HW;
      const int tt = p.T > 0 ? n % p.T : 0;
      const unsigned own = (unsigned)((pl + p.HW) * C1 * 2);
      unsigned own2 = 0;
      if (DUAL) {
        const int rem = (ok ? m : m0) - n * p.HW;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        own2 = (unsigned)((n - n0) * frame2_bytes + ((oy * p.stride2) * p.Wi2 + ox * p.stride2) * C2 * 2);
      }
#pragma unroll
      for (int k = 0; k < PPW; ++k) {
        const int g = PPW * wave + k;
        const int c0 = (2 * g + hsel) * 8;
        lds_void *dst = (lds_void *)(lds + b * kBuf + g * kPlane + i * 1024);
        if (DUAL && c0 >= C1) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX2, dst, 16, (int)(ok ? own2 + (unsigned)((c0 - C1) * 2) : kInvalid), 0, 0, 0);
        } else {
          unsigned off = own;
          bool valid = ok;
          if (p.T > 0 && c0 < p.fold) { off = own + (unsigned)frame_bytes; valid = ok && tt < p.T - 1; }
          else if (p.T > 0 && c0 < 2 * p.fold) { off = own - (unsigned)frame_bytes; valid = ok && tt > 0; }
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, dst, 16, (int)(valid ? off + (unsigned)(c0 * 2) : kInvalid), 0, 0, 0);
        }
      }
    }
  };

  const size_t ybytes = (size_t)p.M * COUT * 2;
  int t = blockIdx.x, nb = 0;
  if (t < ntiles) issue_tile(p.reverse ? ntiles - 1 - t : t, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (; t < ntiles; t += gridDim.x, nb ^= 1) {
    __builtin_amdgcn_s_barrier();
    const int tn = t + gridDim.x;
    if (tn < ntiles) issue_tile(p.reverse ? ntiles - 1 - tn : tn, nb ^ 1);
    const int tt = p.reverse ? ntiles - 1 - t : t;
    const unsigned char *buf = lds + nb * kBuf;
    // output window of the tile (rebased: 32-bit offsets whatever M * COUT is)
    const size_t y0 = (size_t)tt * PX * COUT * 2;
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.y) + y0, 0, (int)(ybytes - y0 > (size_t)PX * COUT * 2 ? (size_t)PX * COUT * 2 : ybytes - y0), 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int pp = mt * 32 + l31;
      const unsigned rd = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));
      f32x16 acc[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const u32x4 px = *reinterpret_cast<const u32x4 *>(buf + rd + g * kPlane);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[nt][g]), __builtin_bit_cast(bf16x8, px),
                                                            acc[nt], 0, 0, 0);
      }
      const unsigned yoff = (unsigned)(pp * COUT * 2);                   // (rows past M fall outside the rebased window)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int ch0 = (wave * NTW + nt) * 32;
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(bias_lds + ch0 + 8 * q + 4 * half);
          pk[q][0] = pack_bf16(fmaxf(acc[nt][4 * q] + b[0], floor_), fmaxf(acc[nt][4 * q + 1] + b[1], floor_));
          pk[q][1] = pack_bf16(fmaxf(acc[nt][4 * q + 2] + b[2], floor_), fmaxf(acc[nt][4 * q + 3] + b[3], floor_));
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
          const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)(yoff + (unsigned)(ch0 * 2 + (2 * half + qq) * 16)), 0, 0);
        }
      }
    }
    wait_vmcnt(MT * NTW * 2);   // the next tile is older than this tile's stores
  }
}

bool conv1x1_wsn_valid(const ConvParams &p) {
  if (p.prec != kPrecBf16 || p.stride != 1 || p.pad != 0 || p.Hi != p.Ho || p.Wi != p.Wo || p.res || p.kseg_len != 0) return false;
  if ((double)(128 + 2.0 * p.Hi * p.Wi) * p.C * 2.0 >= 2.0e9) return false;
  if (p.x2) {   // conv3 + downsample of layer1.0: 64 + 64 -> 256
    return p.T == 0 && p.C == 64 && p.C2 == 64 && p.K1 == 64 && p.Kp == 128 && p.Cout == 256 &&
           (double)(128.0 / (p.Hi * p.Wi) + 2.0) * p.Hi2 * p.Wi2 * p.C2 * 2.0 < 2.0e9;
  }
  if (p.Kp != p.C) return false;
  if (p.T > 0 && (p.N % p.T != 0 || p.fold % 8 != 0 || 2 * p.fold > p.C)) return false;
  return (p.C == 256 && p.Cout == 128) || (p.C == 512 && p.Cout == 128) || (p.C == 512 && p.Cout == 256);
}

bool conv3x3_ws_valid(const ConvParams &p) {
  int tr, tc;
  return p.prec == kPrecBf16 && p.C == 64 && p.Cout == 64 && p.Kp == 576 && p.stride == 1 && p.pad == 1 && p.Hi == p.Ho &&
         p.Wi == p.Wo && !p.res && !p.x2 && p.T == 0 && p.kseg_len == 0 && (double)p.M * 128.0 < 2.0e9 &&
         ws_tile_geometry(p.Hi, p.Wi, &tr, &tc);
}

// ---------------------------------------------------------------------------------------------
// bneck_ws: a WHOLE Bottleneck of layer1 in bf16 as ONE launch -- temporal shift -> conv1 (1x1, CIN -> 64) -> conv2 (3x3,
// 64 -> 64) -> conv3 (1x1, 64 -> 256) -> + identity -> ReLU.  Two forms:
//   CIN = 256  layer1.1 / layer1.2: the identity is the block input itself (residual add in conv3's epilogue);
//   CIN = 64   layer1.0: the identity is the downsample branch, a 1x1 conv of the block input -- as in the engine's fused
//              conv3 + downsample GEMM it is K-concatenated behind conv3 (K = [64 mid | 64 input channels], one packed
//              weight matrix [256][128], one bias), so the block input enters conv3 as a second B operand.
//
// Why: after conv3x3_ws_kernel<true> a layer1 block is two launches that each sit on the HBM roofline (conv1: 2.7 GB at
// 5.2 TB/s; conv2 + conv3: 4.8 GB at 4.7 TB/s): the only lever left is bytes.  Here the 64-channel tensor between
// conv1 and conv2 never exists in memory either, and the block input is streamed ONCE for conv1; its second use (the
// residual / the downsample operand) re-reads rows this CU fetched one or two steps earlier (L2 / Infinity-Cache
// resident) instead of a tensor last touched a launch ago.  Algorithmic HBM bytes per frame: read H*W*CIN*2 + write
// H*W*512 (the separate launches: 3.5x that for CIN = 256).
//
// Structure (one persistent workgroup of four waves per CU, one wave per SIMD, WHOLE FRAMES per workgroup):
//   * a frame is walked top to bottom in steps of two rows.  Step s computes conv1 for rows 2s, 2s + 1 into a LINE
//     BUFFER of four rows in LDS (row r lives in slot (r + 2) & 3; columns 0 and W + 1 and the rows above / below the
//     frame are zeros = conv2's padding), then conv2 + conv3 for output rows 2s - 1, 2s, which need exactly the four
//     buffered rows 2s - 2 .. 2s + 1.  conv1 is computed once per pixel: no halo recompute, no halo re-read.
//   * weights live in registers, DISTRIBUTED over the waves: W1 whole (every wave multiplies its own 32 pixels by all 64
//     mid channels), W2's 32-output-channel slice nt = wave & 1 (36 fragments); W3's 64-output-channel slice of the
//     wave is parked in LDS between steps (its registers are the conv2 phase's pixel fragments).
//   * conv1: the step's 2W pixels are consecutive in memory (full-width rows), 32 per wave.  A wave's 32 pixels x CIN
//     channels arrive by LDS-DMA in a wave-private slot (CIN / 16 planes of 32-byte pixel halves, conv1x1_ws's layout;
//     the temporal shift is the choice of source frame per 16-byte chunk, zeros at the clip's ends), fetched a whole
//     step ahead: 64 KB (16 KB) of the next step's input are in flight per CU while this step computes.  No workgroup
//     barrier inside the phase.
//   * conv2: wave (nt, h) multiplies M-tiles 2h, 2h + 1 of the step's (up to) 128 output pixels by its W2 slice -- one
//     pixel-fragment read per MFMA straight from the line buffer, conv3x3_ws128's loop -- and writes its 32 mid
//     channels to a 16-KB LDS tile in B-fragment order (the K of conv3 is split over the wave pair, so the mid tensor
//     crosses LDS once; it never leaves the CU).
//   * conv3: wave w multiplies all four M-tiles by its 64 output channels; the identity operand comes straight from
//     global memory into registers, prefetched at the top of the step: CIN = 256 -- 16-byte groups of the residual,
//     un-swapped with v_permlane32_swap into the accumulator layout, bias + residual + ReLU + bf16 in
//     conv3x3_ws_kernel<true>'s arithmetic; CIN = 64 -- the input pixel's four B fragments, multiplied by the second
//     half of the packed weights behind the mid tensor, bias + ReLU + bf16 as conv1x1_wsn<.., DUAL>.  16-byte stores.
//   * two barriers per step; every vector-memory wait is a counted vmcnt over a fixed issue order per step
//     [16 identity loads | the LDS-DMA of the next step's input | 16 stores].
// Products enter every accumulator in the separate kernels' order (conv1: k16 groups ascending; conv2: taps, then k16
// groups; conv3: k16 groups, mid before input) and the three epilogues are theirs: bit-identical to the launches it replaces.
// Needs W <= 64 (a row of the line buffer), CMID = 64, fold = CIN / 8.
// ---------------------------------------------------------------------------------------------
constexpr int kBnRP = 66;                          // line-buffer row pitch in pixels: W + 2 <= 66
constexpr int kBnT1Plane = 4 * kBnRP * 32;         // one k16 group of the four buffered rows
constexpr int kBnT1Bytes = 4 * kBnT1Plane;         // 33 792 B
constexpr int kBnMidPlane = 128 * 32;
constexpr int kBnMidOff = kBnT1Bytes;
constexpr int kBnXOff = kBnMidOff + 4 * kBnMidPlane;            // four wave-private input slots
template <int CIN> struct BnLds {
  static constexpr int kSlot = 32 * CIN * 2;                    // 32 pixels x CIN channels: 16 KB / 4 KB
  static constexpr int kW3Off = kBnXOff + 4 * kSlot;            // conv3's weights in fragment order: [it * NG3 + g][lane] 16 B
  static constexpr int kNG3 = CIN == 64 ? 8 : 4;                // k16 groups of conv3's K (the downsample form: mid + input)
  static constexpr int kBiasOff = kW3Off + 8 * kNG3 * 1024;
  static constexpr int kBytes = kBiasOff + (64 + 64 + 256) * 4; // 150 016 B (CIN 256) / 133 632 B (CIN 64)
};

// SHIFT: the temporal shift of conv1's input, fold = CIN / 8 channels from frame t + 1 and as many from t - 1 (the bf16
// formats take shift_div 8 only) -- compile-time, so that a chunk's source row is a register choice and not a table lookup.
template <int CIN, bool SHIFT>
__global__ void __launch_bounds__(256, 1) bneck_ws_kernel(const BneckParams p) {
  constexpr bool DUAL = CIN == 64;
  constexpr int NG1 = CIN / 16;                    // k16 groups of conv1 = planes of an input slot
  constexpr int NG3 = BnLds<CIN>::kNG3;
  constexpr int XROW = CIN * 2;                    // bytes per input pixel
  constexpr int kSlot = BnLds<CIN>::kSlot, kW3Off = BnLds<CIN>::kW3Off, kBiasOff = BnLds<CIN>::kBiasOff;
  constexpr int NDMA = NG1;                        // LDS-DMA operations per step
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int nt2 = wave & 1, hh = wave >> 1;
  const int H = p.H, W = p.W, W2 = 2 * W;
  const int xframe = H * W * XROW, yframe = H * W * 512;
  const int nsteps = H / 2 + 1;

  // ---- the stationary operands
  const __amdgpu_buffer_rsrc_t rsrcW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, 64 * CIN * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, 64 * 576 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w3), 0, 256 * 16 * NG3 * 2, 0x00020000);
  u32x4 w1r[2][NG1], w2r[36];
#pragma unroll
  for (int s = 0; s < 36; ++s)
    w2r[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW2, ((nt2 * 32 + l31) * 576 + s * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < NG1; ++g)
      w1r[nt][g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW1, ((nt * 32 + l31) * CIN + g * 16 + half * 8) * 2, 0, 0);
  // W3 (2 x NG3 fragments per wave) stays in LDS: a step reads it once, into registers that are free in the conv3 phase
#pragma unroll
  for (int k = 0; k < 2 * NG3; ++k) {
    const int fr = wave * 2 * NG3 + k, it = fr / NG3, g = fr - it * NG3;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrcW3, ((32 * it + l31) * (16 * NG3) + 16 * g + 8 * half) * 2, 0, 0);
    *reinterpret_cast<u32x4 *>(lds + kW3Off + fr * 1024 + lane * 16) = v;
  }
#pragma unroll
  for (int s = 0; s < 36; ++s) asm volatile("" : "+a"(w2r[s]));
#pragma unroll
  for (int g = 0; g < NG1; ++g) asm volatile("" : "+a"(w1r[0][g]));

  float *bias1_lds = reinterpret_cast<float *>(lds + kBiasOff);
  float *bias2_lds = bias1_lds + 64, *bias3_lds = bias1_lds + 128;
  if (tid < 64) {
    bias1_lds[tid] = p.bias1[tid];
    bias2_lds[tid] = p.bias2[tid];
  }
  bias3_lds[tid] = p.bias3[tid];
  for (int i = tid; i < kBnT1Bytes / 16; i += 256) *reinterpret_cast<u32x4 *>(lds + i * 16) = u32x4{0u, 0u, 0u, 0u};

  // ---- lane constants
  unsigned char *xs = lds + kBnXOff + wave * kSlot;                     // this wave's input slot
  const unsigned xrd = (unsigned)(l31 * 32 + ((half ^ ((l31 >> 3) & 1)) << 4));   // fragment read: pixel l31 of the slot
  const int pd = lane >> 1;                                             // loader: pixel of the slot, half (lane & 1)
  const int hsel = (lane & 1) ^ ((lane >> 4) & 1);
  const int md = 32 * wave + pd;                                        // ... its index among the step's 2W pixels
  // conv1: this lane's pixel of the step (M-tile = wave)
  const int m1 = 32 * wave + l31;
  const int dr1 = m1 >= W ? 1 : 0, c1 = m1 - dr1 * W;
  const bool ok1 = m1 < W2;

  // LDS-DMA of the input of step s of frame f into this wave's slot: always NDMA operations (dead ones fetch nothing)
  auto issue_x = [&](int f, int s, bool live) {
    const int tt = p.T > 0 ? f % p.T : 0;
    // the descriptor starts one frame BEFORE f (only ever addressed there when frame t - 1 exists)
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + ((long)f - 1) * xframe), 0, 3 * xframe, 0x00020000);
    const int pix = 2 * s * W + md;
    const bool okp = live && md < W2 && pix < H * W;
    // three source rows per pixel: its own frame, frame t + 1, frame t - 1; a k16 group's 32 bytes are the instruction's
    // immediate offset, so a step costs three address registers, not one per group
    const unsigned own = (unsigned)xframe + (unsigned)pix * (unsigned)XROW + (unsigned)hsel * 16u;
    const unsigned vC = okp ? own : kInvalid;
    const unsigned vA = (okp && tt < p.T - 1) ? own + (unsigned)xframe : kInvalid;
    const unsigned vB = (okp && tt > 0) ? own - (unsigned)xframe : kInvalid;
    // fold = CIN / 8 channels: CIN 256 -- k16 groups 0, 1 from t + 1 and 2, 3 from t - 1; CIN 64 -- ONE 16-byte chunk each:
    // the two halves of k16 group 0 (the loader lane's hsel picks the chunk)
    const unsigned v0 = !SHIFT ? vC : (DUAL ? (hsel ? vB : vA) : vA);
    static_for<NG1>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      const unsigned v = !SHIFT ? vC : (DUAL ? (g == 0 ? v0 : vC) : (g < 2 ? vA : (g < 4 ? vB : vC)));
      // (the instruction's immediate offset is added to the LDS address as well as to the memory address: the LDS base
      //  carries g * 1024 - g * 32 so that plane g still starts at g * 1024)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(xs + g * (1024 - 32)), 16, (int)v, 0, g * 32, 0);
    });
  };

  int fi = blockIdx.x;
  if (fi < p.N) issue_x(p.reverse ? p.N - 1 - fi : fi, 0, true);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                         // biases and the zeroed line buffer are in place

  for (; fi < p.N; fi += gridDim.x) {
    const int f = p.reverse ? p.N - 1 - fi : fi;
    const int fnext = fi + (int)gridDim.x < p.N ? (p.reverse ? p.N - 1 - (fi + (int)gridDim.x) : fi + (int)gridDim.x) : -1;
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)f * xframe), 0, xframe, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.y) + (size_t)f * yframe, 0, yframe, 0x00020000);
    // rows -2 and -1 of the new frame (slots 0, 1) are zeros; every wave is past the last conv2 of the previous frame
    for (int i = tid; i < 2 * kBnRP * 2 * 4; i += 256) {                // 2 slots x RP pixels x 2 halves, 4 planes
      const int pl = i / (2 * kBnRP * 2), r = i - pl * (2 * kBnRP * 2);
      *reinterpret_cast<u32x4 *>(lds + pl * kBnT1Plane + r * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    for (int s = 0; s < nsteps; ++s) {
      const int r0 = 2 * s - 1;                                         // output rows r0, r0 + 1; conv1 rows 2s, 2s + 1
      // ================= conv1: rows 2s, 2s + 1 -> line buffer =================
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                 // this step's input has landed (the 16 youngest operations are the previous step's stores)
      // The identity operand of this step's four M-tiles, 4 x 16 bytes per lane and M-tile:
      //   CIN 256  res[mt][2 itl + qq] = bytes [64 it + 16 (2 half + qq), + 16) of the pixel's 512 (the store layout);
      //   CIN 64   res[mt][g] = channels 16 g + 8 half .. + 8 of the pixel: the B fragment of k16 group g.
      // Issue order of a step's vector-memory operations: [8 identity loads, M-tiles 0-1 (conv1 phase) | NDMA LDS-DMA of
      // the next step's input | 8 identity loads, M-tiles 2-3 (top of the conv2 phase) | 16 stores] -- every wait below counts on it.
      u32x4 res[4][4];
      auto issue_res = [&](int mt) {
        const int m = 32 * mt + l31;
        const int dr = m >= W ? 1 : 0, c = m - dr * W, r = r0 + dr;
        const bool ok = m < W2 && (unsigned)r < (unsigned)H;
        if constexpr (DUAL) {
          const unsigned o = ok ? (unsigned)((r * W + c) * XROW + half * 16) : kInvalid;
#pragma unroll
          for (int g = 0; g < 4; ++g) res[mt][g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)o, g * 32, 0);
        } else {
          const unsigned o = ok ? (unsigned)((r * W + c) * 512 + (2 * wave) * 64 + 2 * half * 16) : kInvalid;
#pragma unroll
          for (int k = 0; k < 4; ++k)      // (the constant part rides in the scalar offset: one address register per M-tile)
            res[mt][k] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)o, (k >> 1) * 64 + (k & 1) * 16, 0);
        }
      };
      {
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
        constexpr int GH = NG1 > 8 ? 8 : NG1;                           // the slot's fragments in batches of (at most) eight k16 groups
        u32x4 xf[GH];
#pragma unroll
        for (int gh = 0; gh < NG1 / GH; ++gh) {
#pragma unroll
          for (int g = 0; g < GH; ++g) xf[g] = *reinterpret_cast<const u32x4 *>(xs + xrd + (GH * gh + g) * 1024);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (gh == NG1 / GH - 1) {
            issue_res(0);
            issue_res(1);
            // the slot is free (its fragments are in registers): fetch the next step's input, a whole step ahead
            if (s + 1 < nsteps) issue_x(f, s + 1, true);
            else issue_x(fnext < 0 ? f : fnext, 0, fnext >= 0);
          }
#pragma unroll
          for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[nt][GH * gh + g]), __builtin_bit_cast(bf16x8, xf[g]), acc[nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // bias1, ReLU, bf16; the swap pairs groups (0, 1) and (2, 3): this lane then holds channels 16 g' + 8 half .. + 8 of
        // k16 group g' = 2 nt + qq of its pixel = one 16-byte half of the pixel's entry in plane g' of the line buffer
        const int row = 2 * s + dr1;
        const int pp = ((row + 2) & 3) * kBnRP + c1 + 1;
        const unsigned wr = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));
        const bool inside = row < H;                                    // rows below the frame are conv2's zero padding
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          unsigned pk[4][2];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias1_lds + nt * 32 + 8 * q + 4 * half);
            pk[q][0] = pack_bf16(fmaxf(acc[nt][4 * q] + b[0], 0.f), fmaxf(acc[nt][4 * q + 1] + b[1], 0.f));
            pk[q][1] = pack_bf16(fmaxf(acc[nt][4 * q + 2] + b[2], 0.f), fmaxf(acc[nt][4 * q + 3] + b[3], 0.f));
          }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][0], pk[2 * qq + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][1], pk[2 * qq + 1][1], false, false);
            const u32x4 v = inside ? u32x4{s0[0], s1[0], s0[1], s1[1]} : u32x4{0u, 0u, 0u, 0u};
            if (ok1) *reinterpret_cast<u32x4 *>(lds + (2 * nt + qq) * kBnT1Plane + wr) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();    // rows 2s - 2 .. 2s + 1 are complete; nobody still reads the mid tile of the previous step
      // ================= conv2: output rows r0, r0 + 1, M-tiles 2 hh, 2 hh + 1, mid channels 32 nt2 .. + 32 =================
      issue_res(2);
      issue_res(3);
      {
        int cc[2], drr[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int ml = 32 * (2 * hh + m) + l31;
          const bool ok = ml < W2;
          drr[m] = (ok && ml >= W) ? 1 : 0;
          cc[m] = ok ? ml - drr[m] * W : 0;
        }
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
        u32x4 px[4][2];
        unsigned tb[2] = {0u, 0u};
        auto rd = [&](int st) {
          const int tap = st >> 2, g = st & 3, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (g == 0) {
              const int pp = ((2 * s + drr[m] + ky) & 3) * kBnRP + cc[m] + kx;      // row r0 + dr - 1 + ky, column c - 1 + kx (+ 1)
              tb[m] = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));
            }
            px[st & 3][m] = *reinterpret_cast<const u32x4 *>(lds + tb[m] + g * kBnT1Plane);
          }
        };
        rd(0); rd(1); rd(2);
        static_for<36>([&](auto sc) __attribute__((always_inline)) {
          constexpr int st = decltype(sc)::value;
          if constexpr (st + 3 < 36) rd(st + 3);
#pragma unroll
          for (int m = 0; m < 2; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2r[st]), __builtin_bit_cast(bf16x8, px[st & 3][m]),
                                                             acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
        // bias2, ReLU, bf16 -> the mid tile, in conv3's B-fragment order (plane g' = 2 nt2 + qq)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int ml = 32 * (2 * hh + m) + l31;
          const unsigned wr = (unsigned)(ml * 32 + ((half ^ ((ml >> 3) & 1)) << 4));
          unsigned pk[4][2];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias2_lds + nt2 * 32 + 8 * q + 4 * half);
            pk[q][0] = pack_bf16(fmaxf(acc[m][4 * q] + b[0], 0.f), fmaxf(acc[m][4 * q + 1] + b[1], 0.f));
            pk[q][1] = pack_bf16(fmaxf(acc[m][4 * q + 2] + b[2], 0.f), fmaxf(acc[m][4 * q + 3] + b[3], 0.f));
          }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][0], pk[2 * qq + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][1], pk[2 * qq + 1][1], false, false);
            *reinterpret_cast<u32x4 *>(lds + kBnMidOff + (2 * nt2 + qq) * kBnMidPlane + wr) = u32x4{s0[0], s1[0], s0[1], s1[1]};
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();    // the mid tile is complete; the line buffer may be overwritten by the next step
      // ================= conv3 + identity: all four M-tiles, output channels 64 wave .. + 64 =================
      u32x4 w3r[2][NG3];
#pragma unroll
      for (int itl = 0; itl < 2; ++itl)
#pragma unroll
        for (int g = 0; g < NG3; ++g)
          w3r[itl][g] = *reinterpret_cast<const u32x4 *>(lds + kW3Off + ((2 * wave + itl) * NG3 + g) * 1024 + lane * 16);
      static_for<4>([&](auto mc) __attribute__((always_inline)) {
        constexpr int mt = decltype(mc)::value;
        const int ml = 32 * mt + l31;
        const unsigned mrd = (unsigned)(ml * 32 + ((half ^ ((ml >> 3) & 1)) << 4));
        u32x4 bf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bf[g] = *reinterpret_cast<const u32x4 *>(lds + kBnMidOff + g * kBnMidPlane + mrd);
        f32x16 c3[2];
#pragma unroll
        for (int itl = 0; itl < 2; ++itl)
#pragma unroll
          for (int e = 0; e < 16; ++e) c3[itl][e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int itl = 0; itl < 2; ++itl)
            c3[itl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[itl][g]), __builtin_bit_cast(bf16x8, bf[g]), c3[itl], 0, 0, 0);
        const int dr = ml >= W ? 1 : 0, c = ml - dr * W, r = r0 + dr;
        const unsigned yo = (ml < W2 && (unsigned)r < (unsigned)H) ? (unsigned)((r * W + c) * 512 + (2 * wave) * 64 + 2 * half * 16) : kInvalid;
        // this M-tile's identity operand; younger operations: M-tiles 0, 1 -- the other early loads (4 / 0), the next input
        // (NDMA), the late loads (8), the stores so far (0 / 4) = NDMA + 12; M-tiles 2, 3 -- the other late loads and the
        // stores so far = 12
        if constexpr (mt < 2) wait_vmcnt(NDMA + 12);
        else wait_vmcnt(12);
        if constexpr (DUAL) {   // the downsample branch: K continues over the block input's 64 channels
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int itl = 0; itl < 2; ++itl)
              c3[itl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[itl][4 + g]), __builtin_bit_cast(bf16x8, res[mt][g]), c3[itl], 0, 0, 0);
        }
#pragma unroll
        for (int itl = 0; itl < 2; ++itl) {
          unsigned pk[4][2];
          if constexpr (DUAL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 b = *reinterpret_cast<const f32x4 *>(bias3_lds + (2 * wave + itl) * 32 + 8 * q + 4 * half);
              pk[q][0] = pack_bf16(fmaxf(c3[itl][4 * q] + b[0], 0.f), fmaxf(c3[itl][4 * q + 1] + b[1], 0.f));
              pk[q][1] = pack_bf16(fmaxf(c3[itl][4 * q + 2] + b[2], 0.f), fmaxf(c3[itl][4 * q + 3] + b[3], 0.f));
            }
          } else {
            // the residual's 16-byte groups -> accumulator layout: the store swap backwards
            unsigned rp[4][2];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
              for (int w2 = 0; w2 < 2; ++w2) {
                const auto sw = __builtin_amdgcn_permlane32_swap(res[mt][2 * itl + qq][w2], res[mt][2 * itl + qq][2 + w2], false, false);
                rp[qq][w2] = sw[0];
                rp[qq + 2][w2] = sw[1];
              }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 b = *reinterpret_cast<const f32x4 *>(bias3_lds + (2 * wave + itl) * 32 + 8 * q + 4 * half);
#pragma unroll
              for (int w2 = 0; w2 < 2; ++w2) {
                const unsigned rw = rp[q][w2];
                f32x2 v = f32x2{c3[itl][4 * q + 2 * w2], c3[itl][4 * q + 2 * w2 + 1]} + f32x2{b[2 * w2], b[2 * w2 + 1]};
                v += f32x2{__builtin_bit_cast(float, rw << 16), __builtin_bit_cast(float, rw & 0xFFFF0000u)};
                pk[q][w2] = pack_bf16(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
              }
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
            const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)yo, itl * 64 + qq * 16, 0);
          }
        }
      });
    }
  }
}

bool bneck_ws_valid(int cin, int n, int h, int w, int T, int fold) {
  return (cin == 256 || cin == 64) && n > 0 && h > 0 && w >= 1 && w <= 64 && (double)h * w * 512.0 * 3.0 < 2.0e9 &&
         (T == 0 || (T > 0 && n % T == 0 && fold == cin / 8));
}

hipError_t launch_bneck_ws(const BneckParams &p, hipStream_t s) {
  if (!p.x || !p.w1 || !p.bias1 || !p.w2 || !p.bias2 || !p.w3 || !p.bias3 || !p.y) return hipErrorInvalidValue;
  if (!bneck_ws_valid(p.cin, p.N, p.H, p.W, p.T, p.fold)) return hipErrorInvalidValue;
  const DeviceInfo &di = device_info();
  if (di.status != hipSuccess) return di.status;
  const dim3 grid((unsigned)(p.N < di.n_cu ? p.N : di.n_cu)), block(256);
  if (p.cin == 256) {
    if (p.T > 0) hipLaunchKernelGGL((bneck_ws_kernel<256, true>), grid, block, BnLds<256>::kBytes, s, p);
    else hipLaunchKernelGGL((bneck_ws_kernel<256, false>), grid, block, BnLds<256>::kBytes, s, p);
  } else {
    if (p.T > 0) hipLaunchKernelGGL((bneck_ws_kernel<64, true>), grid, block, BnLds<64>::kBytes, s, p);
    else hipLaunchKernelGGL((bneck_ws_kernel<64, false>), grid, block, BnLds<64>::kBytes, s, p);
  }
  return hipGetLastError();
}

static const DeviceInfo &device_info() {
  static std::mutex mu;
  static std::map<int, DeviceInfo> seen;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  auto it = seen.find(dev);
  if (it != seen.end()) return it->second;
  DeviceInfo di;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) di.n_cu = prop.multiProcessorCount;
  auto opt_in = [&](const void *fn, size_t bytes) {
    const hipError_t st = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (st != hipSuccess && di.status == hipSuccess) di.status = st;
  };
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, false>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, true>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<3, false>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, false, true, false>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, false>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, true>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<3, false>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, false, true, false>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256p_kernel<1, false, false, true>), kLds256pBytes);
  opt_in(reinterpret_cast<const void *>(&conv_bf16_256_kernel<1, false, false, true>), kLds256Bytes);
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws_kernel<false>), kWsLdsBytes);
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws_kernel<true>), kWsLdsBytes3All);
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws128_kernel), kW8LdsBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<256, true>), BnLds<256>::kBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<256, false>), BnLds<256>::kBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<64, true>), BnLds<64>::kBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<64, false>), BnLds<64>::kBytes);
  opt_in(reinterpret_cast<const void *>(&conv1x1_ws_kernel<256>), 2 * 16 * 4096 + 256);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<128, 256, true>), 2 * 65536 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<256, 128, false>), 2 * 65536 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<512, 128, false>), 2 * 65536 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<512, 256, false>), 2 * 65536 + 1024);
  return seen.emplace(dev, di).first->second;
}

static int ws_grid_setup() { return device_info().n_cu; }

static hipError_t launch_conv3x3_ws(ConvParams p, hipStream_t s) {
  if (conv3x3_ws128_valid(p)) {
    WsParams q{};
    q.x = p.x; q.w2 = p.w; q.bias2 = p.bias; q.y = p.y;
    q.N = p.N; q.H = p.Hi; q.W = p.Wi; q.M = p.M; q.relu = p.relu; q.reverse = p.reverse;
    ws_tile_geometry(q.H, q.W, &q.tr, &q.tc, 128, kW8PatchMax);
    const long ntiles = (long)q.N * ((q.H + q.tr - 1) / q.tr) * ((q.W + q.tc - 1) / q.tc);
    const int n_cu = ws_grid_setup();
    if (device_info().status != hipSuccess) return device_info().status;
    hipLaunchKernelGGL(conv3x3_ws128_kernel, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kW8LdsBytes, s, q);
    return hipGetLastError();
  }
  if (!conv3x3_ws_valid(p)) return hipErrorInvalidValue;
  WsParams q{};
  q.x = p.x; q.w2 = p.w; q.bias2 = p.bias; q.y = p.y;
  q.N = p.N; q.H = p.Hi; q.W = p.Wi; q.M = p.M; q.relu = p.relu; q.reverse = p.reverse;
  ws_tile_geometry(q.H, q.W, &q.tr, &q.tc);
  const long ntiles = (long)q.N * ((q.H + q.tr - 1) / q.tr) * ((q.W + q.tc - 1) / q.tc);
  const int n_cu = ws_grid_setup();
  if (device_info().status != hipSuccess) return device_info().status;
  hipLaunchKernelGGL(conv3x3_ws_kernel<false>, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kWsLdsBytes, s, q);
  return hipGetLastError();
}

static hipError_t launch_conv1x1_ws(const ConvParams &p, hipStream_t s) {
  if (!conv1x1_ws_valid(p)) return hipErrorInvalidValue;
  Ws1Params q{};
  q.x = p.x; q.w = p.w; q.bias = p.bias; q.y = p.y;
  q.M = p.M; q.HW = p.Hi * p.Wi; q.T = p.T; q.fold = p.fold; q.relu = p.relu; q.reverse = p.reverse;
  const int n_cu = ws_grid_setup();
  if (device_info().status != hipSuccess) return device_info().status;
  const int ntiles = (p.M + 127) / 128;
  const unsigned grid = (unsigned)(ntiles < n_cu ? ntiles : n_cu);
  if (p.C == 256) hipLaunchKernelGGL(conv1x1_ws_kernel<256>, dim3(grid), dim3(256), 2 * 16 * 4096 + 256, s, q);
  else hipLaunchKernelGGL(conv1x1_ws_kernel<64>, dim3(grid), dim3(256), 2 * 4 * 4096 + 256, s, q);
  return hipGetLastError();
}

static hipError_t launch_conv1x1_wsn(const ConvParams &p, hipStream_t s) {
  if (!conv1x1_wsn_valid(p)) return hipErrorInvalidValue;
  WsnParams q{};
  q.x = p.x; q.x2 = p.x2; q.w = p.w; q.bias = p.bias; q.y = p.y;
  q.M = p.M; q.HW = p.Hi * p.Wi; q.Wo = p.Wo; q.T = p.T; q.fold = p.fold; q.relu = p.relu; q.reverse = p.reverse;
  q.K1 = p.x2 ? p.K1 : p.C; q.Hi2 = p.Hi2; q.Wi2 = p.Wi2; q.stride2 = p.stride2;
  const int n_cu = ws_grid_setup();
  if (device_info().status != hipSuccess) return device_info().status;
  const int px = p.Kp <= 256 ? 128 : 64;
  const int ntiles = (p.M + px - 1) / px;
  const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(256);
  constexpr size_t kLds = 2 * 65536 + 1024;
  if (p.x2) hipLaunchKernelGGL((conv1x1_wsn_kernel<128, 256, true>), grid, block, kLds, s, q);
  else if (p.C == 256) hipLaunchKernelGGL((conv1x1_wsn_kernel<256, 128, false>), grid, block, kLds, s, q);
  else if (p.Cout == 128) hipLaunchKernelGGL((conv1x1_wsn_kernel<512, 128, false>), grid, block, kLds, s, q);
  else hipLaunchKernelGGL((conv1x1_wsn_kernel<512, 256, false>), grid, block, kLds, s, q);
  return hipGetLastError();
}

// Fused23Params with bf16 operands: w3f = conv3's packed weights [256][64] bf16 (row-major, as launch_conv takes them).
bool conv23_ws_valid(int n, int h, int w) {
  int tr, tc;
  return n > 0 && h > 0 && w > 0 && (double)h * w * 512.0 < 2.0e9 && (double)n * h * w < 2.0e9 && ws_tile_geometry(h, w, &tr, &tc);
}

static hipError_t launch_conv23_ws(const Fused23Params &p, hipStream_t s) {
  if (!conv23_ws_valid(p.N, p.H, p.W) || p.kseg_len != 0) return hipErrorInvalidValue;
  WsParams q{};
  q.x = p.x; q.w2 = p.w2; q.bias2 = p.bias2; q.w3 = p.w3f; q.bias3 = p.bias3; q.res = p.res; q.y = p.y;
  q.N = p.N; q.H = p.H; q.W = p.W; q.M = p.M; q.relu = 1; q.reverse = p.reverse;
  ws_tile_geometry(q.H, q.W, &q.tr, &q.tc);
  const long ntiles = (long)q.N * ((q.H + q.tr - 1) / q.tr) * ((q.W + q.tc - 1) / q.tc);
  const int n_cu = ws_grid_setup();
  if (device_info().status != hipSuccess) return device_info().status;
  hipLaunchKernelGGL(conv3x3_ws_kernel<true>, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kWsLdsBytes3All, s, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conv23_fused: Bottleneck.conv2 (3x3, stride 1) + bn2 + ReLU + conv3 (1x1) + bn3 + residual + ReLU in ONE kernel,
// for the blocks whose mid tensor is the whole K of conv3 (CMID = 64: layer1, CMID = 128: layer2; fp32 and split-bf16).
//
// A workgroup owns 64 output pixels.  Phase A is the 3x3 implicit GEMM of conv_igemm's fp32 64x64 pipeline
// (register-resident K-step, one LDS buffer, segmented K where the layer is segmented) over ALL CMID output
// channels: 2 x (CMID / 32) waves, one 32x32 accumulator tile each.  Phase B turns the accumulators into the
// tensor the un-fused path would have stored -- relu(acc + bias2) -- but keeps it in LDS ([64][CMID + 4] fp32).
// Phase C multiplies that tile by W3 in four chunks of BNC = CMID output channels: the A fragments come from the
// LDS tile, the B fragments straight from global memory (W3 is pre-packed on the host in fragment order, so a
// wave's fragment load is one fully coalesced 1-KiB read of an L2-resident 64 / 256 KB matrix: no LDS staging for
// W3), and every 32 x 32 accumulator tile leaves through a wave-private LDS slab (no workgroup barrier in the whole
// phase) with conv_igemm's epilogue arithmetic (+ bias3, + residual, ReLU, 16-byte stores of whole row segments).
// The CMID-channel mid tensor -- 205 MB per layer1 block at batch 32 -- is never written or re-read, and the
// HBM-bound (layer1) / prologue-bound (layer2, K = 128) conv3 launch disappears.
//
// Every product enters its accumulator in the same order as in the two separate kernels (same k order, same
// segment sums, same epilogue arithmetic), so the output is bit-identical to them.
// ---------------------------------------------------------------------------------------------
template <int CMID, bool X3>
// (second argument = waves per SIMD: 16 waves per CU in both geometries, which caps the allocation at 128 registers)
__global__ void __launch_bounds__(128 * (CMID / 32), 4) conv23_fused_kernel(const Fused23Params p) {
  constexpr int WGN = CMID / 32;         // waves along the CMID channels (phase A) / along a chunk (phase C)
  constexpr int NT = 128 * WGN;          // 2 x WGN waves
  constexpr int LRP = NT / 8;            // loader rows per pass
  constexpr int APASS = 64 / LRP, BPASS = CMID / LRP;
  constexpr int NITEMS = APASS + BPASS;
  constexpr int TLD = CMID + 4;          // row stride of the mid tile / of the epilogue staging (floats)
  constexpr int BNC = CMID;              // output channels per phase-C chunk
  constexpr int NCHUNK = 4;              // Cout3 = 4 * CMID
  constexpr int R0a = (64 + CMID) * kLds > 64 * TLD ? (64 + CMID) * kLds : 64 * TLD;
  constexpr int R0 = R0a > (NT / 64) * 32 * 36 ? R0a : (NT / 64) * 32 * 36;   // staging buffer | phase-B tile | one 32 x 36 slab per wave
  __shared__ __attribute__((aligned(16))) float smem[R0 + 64 * TLD];
  float *Ts = smem + R0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l31 = lane & 31;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  if (p.reverse) tile = nwg - 1 - tile;
  const int m0 = tile * 64;

  const int HW = p.H * p.W;
  const int frame0 = m0 / HW;
  const int frame_bytes = HW * CMID * 4;
  const size_t a_bytes = ((size_t)p.N - frame0) * frame_bytes;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(reinterpret_cast<const char *>(p.x) + (size_t)frame0 * frame_bytes), 0,
      (int)(a_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : a_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w2), 0, CMID * 9 * CMID * 4, 0x00020000);

  const int lrow = tid >> 3, chunk = tid & 7;
  unsigned a_off[APASS], a_mask[APASS], b_off[BPASS];
#pragma unroll
  for (int pp = 0; pp < APASS; ++pp) {
    const int m = m0 + lrow + LRP * pp;
    const bool ok = m < p.M;
    const int mm = ok ? m : m0;
    const int n = mm / HW, rem = mm - n * HW;
    const int oy = rem / p.W, ox = rem - oy * p.W;
    a_off[pp] = (unsigned)((n - frame0) * frame_bytes + ((oy - 1) * p.W + (ox - 1)) * CMID * 4 + chunk * 16);
    unsigned mask = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        if ((unsigned)(oy - 1 + ky) < (unsigned)p.H && (unsigned)(ox - 1 + kx) < (unsigned)p.W) mask |= 1u << (ky * 3 + kx);
    a_mask[pp] = ok ? mask : 0u;
  }
#pragma unroll
  for (int pp = 0; pp < BPASS; ++pp) b_off[pp] = (unsigned)((lrow + LRP * pp) * 9 * CMID * 4 + chunk * 16);

  f32x4 ra[APASS], rb[BPASS];
  const int nk = 9 * CMID / kBK;
  auto gload_item = [&](int kt, int item) {
    const unsigned dead = (~(unsigned)((kt - nk) >> 31)) & kInvalid;   // K-steps past the end read zeros
    if (item < APASS) {
      const int tap = (kt * kBK) / CMID;
      const int ky = tap / 3, kx = tap - ky * 3;
      const unsigned tap_off = (unsigned)(((ky * p.W + kx) * CMID + (kt * kBK - tap * CMID)) * 4);
      ra[item] = buf_load4(rsrcA, (((a_mask[item] >> tap) & 1u) ? a_off[item] + tap_off : kInvalid) | dead, 0);
    } else {
      rb[item - APASS] = buf_load4(rsrcB, b_off[item - APASS] | dead, (unsigned)kt * (kBK * 4));
    }
  };
  auto lstore_item = [&](int item) {
    if (item < APASS)
      *reinterpret_cast<f32x4 *>(smem + (lrow + LRP * item) * kLds + chunk * 4) = ra[item];
    else
      *reinterpret_cast<f32x4 *>(smem + (64 + lrow + LRP * (item - APASS)) * kLds + chunk * 4) = rb[item - APASS];
  };

  // residual / output window of this workgroup: rows m0 .., all 4 * CMID channels (rows past M are dropped / zero)
  const int cout = NCHUNK * BNC;
  const size_t y_bytes = ((size_t)p.M - m0) * cout * 4;
  const int y_rec = (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes);
  const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.res + (size_t)m0 * cout), 0, y_rec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)m0 * cout, 0, y_rec, 0x00020000);

  // ---- phase A: 3x3 conv, K = 9 * CMID, register-resident K-step pipeline -------------------------------
  f32x16 acc, tot;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = tot[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NITEMS; ++it) gload_item(0, it);
#pragma unroll
  for (int it = 0; it < NITEMS; ++it) lstore_item(it);
#pragma unroll
  for (int it = 0; it < NITEMS; ++it) gload_item(1, it);
  __syncthreads();
  const bool seg = p.kseg_len > 0;
  const int seg_len = seg ? p.kseg_len : 0x3fffffff;
  f32x4 ra_[4], rb_[4];     // fp32: four k-groups of A / B fragments.  split-bf16: [2q] = hi, [2q + 1] = lo of k16 group q
  for (int kt = 0; kt < nk;) {
    const int kend = kt + seg_len < nk ? kt + seg_len : nk;
    for (; kt < kend; ++kt) {
      {
        const float *As = smem + (wm * 32 + l31) * kLds + (X3 ? 0 : half * 4);
        const float *Bs = smem + (64 + wn * 32 + l31) * kLds + (X3 ? 0 : half * 4);
        if constexpr (X3) {   // an LDS row = 4 channel groups [hi x8 | lo x8]; k16 group q reads group 2q + half
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            ra_[2 * q] = *reinterpret_cast<const f32x4 *>(As + (2 * q + half) * 8);
            ra_[2 * q + 1] = *reinterpret_cast<const f32x4 *>(As + (2 * q + half) * 8 + 4);
            rb_[2 * q] = *reinterpret_cast<const f32x4 *>(Bs + (2 * q + half) * 8);
            rb_[2 * q + 1] = *reinterpret_cast<const f32x4 *>(Bs + (2 * q + half) * 8 + 4);
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            ra_[kk] = *reinterpret_cast<const f32x4 *>(As + kk * 8);
            rb_[kk] = *reinterpret_cast<const f32x4 *>(Bs + kk * 8);
          }
        }
      }
      __syncthreads();  // every wave holds its fragments: the buffer may be overwritten
      int cnt = 0;
      if constexpr (X3) {
        // a*b = ah*bh + ah*bl + al*bh, in conv_igemm's order; the 2 * NITEMS loader items ride on the first five MFMAs
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, t == 2 ? ra_[2 * q + 1] : ra_[2 * q]);
            const bf16x8 b = __builtin_bit_cast(bf16x8, t == 1 ? rb_[2 * q + 1] : rb_[2 * q]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            ++cnt;
            const int done = cnt < 5 ? (cnt * 2 * NITEMS) / 5 : 2 * NITEMS;
            const int before = cnt - 1 < 5 ? ((cnt - 1) * 2 * NITEMS) / 5 : 2 * NITEMS;
#pragma unroll
            for (int it = before; it < done; ++it) {
              if (it < NITEMS) lstore_item(it);
              else gload_item(kt + 2, it - NITEMS);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra_[kk][s4], rb_[kk][s4], acc, 0, 0, 0);
            ++cnt;
            const int done = cnt < 12 ? (cnt * 2 * NITEMS) / 12 : 2 * NITEMS;
            const int before = cnt - 1 < 12 ? ((cnt - 1) * 2 * NITEMS) / 12 : 2 * NITEMS;
#pragma unroll
            for (int it = before; it < done; ++it) {
              if (it < NITEMS) lstore_item(it);
              else gload_item(kt + 2, it - NITEMS);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
      }
      __syncthreads();  // tile kt+1 is complete in LDS
    }
    if (seg) {          // out = ((0 + s0) + s1) + ..., exactly as conv_igemm<SEG> sums its segments
      tot += acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    }
  }

  // ---- phase B: the mid tensor tile, as the un-fused conv2 would have stored it, into LDS -----------------
  if constexpr (!X3) {
    const float b2 = p.bias2[wn * 32 + l31];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      Ts[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * TLD + wn * 32 + l31] = fmaxf((seg ? tot[e] : acc[e]) + b2, 0.f);
  } else {
    // split-bf16: through the staging tile, 8 channels per thread, exactly conv_igemm's epilogue (bias, ReLU,
    // hi = bf16(v), lo = bf16(v - hi)) -- but the 32-byte group [hi x8 | lo x8] goes to the LDS tile instead of HBM
    float *Cst = smem;
#pragma unroll
    for (int e = 0; e < 16; ++e) Cst[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * TLD + wn * 32 + l31] = acc[e];
    __syncthreads();
    constexpr int TPRB = CMID / 8, RPPB = NT / TPRB, EPB = 64 / RPPB;
    const int bcol = (tid % TPRB) * 8, brow = tid / TPRB;
    const f32x4 bb0 = *reinterpret_cast<const f32x4 *>(p.bias2 + bcol), bb1 = *reinterpret_cast<const f32x4 *>(p.bias2 + bcol + 4);
#pragma unroll
    for (int k = 0; k < EPB; ++k) {
      const int rr = brow + k * RPPB;
      const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cst + rr * TLD + bcol);
      const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cst + rr * TLD + bcol + 4);
      const float v[8] = {c0[0] + bb0[0], c0[1] + bb0[1], c0[2] + bb0[2], c0[3] + bb0[3],
                          c1[0] + bb1[0], c1[1] + bb1[1], c1[2] + bb1[2], c1[3] + bb1[3]};
      u32x4 oh, ol;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) {
        unsigned hw, lw;
        split_pair(fmaxf(v[2 * w2], 0.f), fmaxf(v[2 * w2 + 1], 0.f), &hw, &lw);
        oh[w2] = hw;
        ol[w2] = lw;
      }
      *reinterpret_cast<u32x4 *>(Ts + rr * TLD + bcol) = oh;
      *reinterpret_cast<u32x4 *>(Ts + rr * TLD + bcol + 4) = ol;
    }
  }
  __syncthreads();

  // ---- phase C: [64 x CMID] x W3^T, NCHUNK chunks of BNC output channels, two chunks at a time -------------
  // Each wave owns the 32 x 32 tile (wm, wn) of every chunk.  Per pair of chunks: the residual rows are requested,
  // the two accumulators run over K = CMID with A fragments from the LDS tile and B fragments from global memory
  // (fragment-ordered W3, prefetched one k-group ahead), then each tile goes through a WAVE-PRIVATE LDS slab
  // (no workgroup barrier anywhere in this phase) to become whole 128-byte row segments: + bias3, + residual, ReLU.
  constexpr int NKK = CMID / 8;          // 16-byte B fragments per chunk and wave (fp32: k-groups of 8; split-bf16: k16 groups x {hi, lo})
  constexpr int CWLD = 36;               // slab row stride (floats): 16-byte aligned rows
  static_assert(R0 >= (NT / 64) * 32 * CWLD, "the staging region holds one 32 x 32 slab per wave");
  float *Cw = smem + wave * 32 * CWLD;
  const f32x4 *w3f = reinterpret_cast<const f32x4 *>(p.w3f);
  const float *Ta = Ts + (wm * 32 + l31) * TLD + (X3 ? half * 8 : half * 4);
  // epilogue mapping inside a 32 x 32 tile: fp32 4 channels per lane (8 lanes per row, 8 rows per pass, 4 passes);
  // split-bf16 8 channels = one 32-byte group per lane (4 lanes per row, 16 rows per pass, 2 passes)
  constexpr int LPR = X3 ? 4 : 8, RPW = 64 / LPR, NPW = 32 / RPW, ECH = 32 / LPR;
  const int er = lane / LPR, ec = (lane % LPR) * ECH;
#pragma unroll 1
  for (int jp = 0; jp < NCHUNK; jp += 2) {
    f32x4 rres[2][NPW], rres2[2][X3 ? NPW : 1];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int k = 0; k < NPW; ++k) {
        const unsigned o = (unsigned)(((wm * 32 + er + k * RPW) * cout + (jp + jj) * BNC + wn * 32 + ec) * 4);
        rres[jj][k] = buf_load4(rsrcR, o, 0);
        if constexpr (X3) rres2[jj][k] = buf_load4(rsrcR, o + 16, 0);
      }
    f32x16 c3[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 16; ++e) c3[jj][e] = 0.f;
    constexpr int FPG = X3 ? 2 : 1;      // fragments per k-group: split-bf16 hi + lo
    constexpr int NG = NKK / FPG;        // k-groups
    f32x4 bcur[2][FPG], bnxt[2][FPG];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int f = 0; f < FPG; ++f) bcur[jj][f] = w3f[(((jp + jj) * WGN + wn) * NKK + f) * 64 + lane];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int gn = g + 1 < NG ? g + 1 : g;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int f = 0; f < FPG; ++f) bnxt[jj][f] = w3f[(((jp + jj) * WGN + wn) * NKK + gn * FPG + f) * 64 + lane];
      if constexpr (X3) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Ta + g * 16));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(Ta + g * 16 + 4));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[jj][0]), bl = __builtin_bit_cast(bf16x8, bcur[jj][1]);
          c3[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c3[jj], 0, 0, 0);
          c3[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c3[jj], 0, 0, 0);
          c3[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c3[jj], 0, 0, 0);
        }
      } else {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(Ta + g * 8);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            c3[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], bcur[jj][0][s4], c3[jj], 0, 0, 0);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int f = 0; f < FPG; ++f) bcur[jj][f] = bnxt[jj][f];
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int col0 = (jp + jj) * BNC + wn * 32 + ec;
#pragma unroll
      for (int e = 0; e < 16; ++e) Cw[((e & 3) + 8 * (e >> 2) + 4 * half) * CWLD + l31] = c3[jj][e];
      // (a wave's LDS operations complete in order: its own reads below see its own writes without a barrier)
      if constexpr (X3) {
        const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(p.bias3 + col0), bias1 = *reinterpret_cast<const f32x4 *>(p.bias3 + col0 + 4);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
          const int rr = er + k * RPW;
          const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cw + rr * CWLD + ec);
          const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cw + rr * CWLD + ec + 4);
          float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                        c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
          const u32x4 rh = __builtin_bit_cast(u32x4, rres[jj][k]), rl = __builtin_bit_cast(u32x4, rres2[jj][k]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += split_elem(rh, e) + split_elem(rl, e);
          u32x4 oh, ol;
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            unsigned hw, lw;
            split_pair(fmaxf(v[2 * w2], 0.f), fmaxf(v[2 * w2 + 1], 0.f), &hw, &lw);
            oh[w2] = hw;
            ol[w2] = lw;
          }
          const int o = ((wm * 32 + rr) * cout + col0) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(oh, rsrcY, o, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(ol, rsrcY, o + 16, 0, 0);
        }
      } else {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias3 + col0);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
          const int rr = er + k * RPW;
          f32x4 v = *reinterpret_cast<const f32x4 *>(Cw + rr * CWLD + ec);
          v += bias;
          v += rres[jj][k];
          v[0] = fmaxf(v[0], 0.f);
          v[1] = fmaxf(v[1], 0.f);
          v[2] = fmaxf(v[2], 0.f);
          v[3] = fmaxf(v[3], 0.f);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcY, (int)(((wm * 32 + rr) * cout + col0) * 4), 0, 0);
        }
      }
    }
  }
}

hipError_t launch_conv23_fused(const Fused23Params &p_in, int cmid, int prec, hipStream_t s) {
  Fused23Params p = p_in;
  if (prec == kPrecBf16) {   // weight-stationary form (conv3x3_ws_kernel<true>): layer1's geometry only
    if (cmid != 64 || !p.x || !p.w2 || !p.bias2 || !p.w3f || !p.bias3 || !p.res || !p.y || p.M != p.N * p.H * p.W) return hipErrorInvalidValue;
    return launch_conv23_ws(p, s);
  }
  if (prec != kPrecF32 && prec != kPrecBf16x3) return hipErrorInvalidValue;
  if (prec == kPrecBf16x3 && p.kseg_len != 0) return hipErrorInvalidValue;   // (only fp32 layers are segmented)
  if (!p.x || !p.w2 || !p.bias2 || !p.w3f || !p.bias3 || !p.res || !p.y) return hipErrorInvalidValue;
  if ((cmid != 64 && cmid != 128) || p.N <= 0 || p.H <= 0 || p.W <= 0 || p.M != p.N * p.H * p.W) return hipErrorInvalidValue;
  if (p.kseg_len < 0 || 6.0 * p.H * p.W * cmid * 4.0 > 2.0e9) return hipErrorInvalidValue;   // 32-bit offsets per window
  const unsigned grid = (unsigned)((p.M + 63) / 64);
  if (prec == kPrecBf16x3) {
    if (cmid == 64) hipLaunchKernelGGL((conv23_fused_kernel<64, true>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv23_fused_kernel<128, true>), dim3(grid), dim3(512), 0, s, p);
  } else {
    if (cmid == 64) hipLaunchKernelGGL((conv23_fused_kernel<64, false>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv23_fused_kernel<128, false>), dim3(grid), dim3(512), 0, s, p);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// stem_direct: the 7x7 stride-2 stem of the bf16 formats as a direct convolution from an LDS-resident input patch.
// The generic implicit-GEMM loader gathers 28 groups of 8 elements per output pixel from L2 (each input pixel pair
// is fetched ~12 times): with 3 input channels that gather, not the MFMA or HBM, bounds the stem in these formats.
// Here a persistent workgroup keeps the packed weights [64][224] in LDS and, per (2 * WAVES) x 16 tile of output
// pixels, loads the (4 * WAVES + 5) x 19 pixel-pair patch it needs ONCE (prefetched into registers under the previous
// tile), then builds every MFMA A fragment straight from that patch: with K ordered (ky, pair j, pixel-in-pair, c4) a
// fragment (8 consecutive k) is exactly one group of the patch at row 2*oy + ky, pair ox + j.  14 k16-steps x 2
// N-tiles per wave and tile, no barrier inside.  Same products in the same order per accumulator as conv_igemm's
// stem (whose trailing all-zero K padding is skipped), so results are bit-identical to it.
// ---------------------------------------------------------------------------------------------
constexpr int kStemTW = 16, kStemPC = kStemTW + 3;  // output tile width; input patch width in pixel pairs (19)

// X3 = false: TSM_DTYPE_BF16 (16-byte groups of 8 bf16);  true: split-bf16 (32-byte groups [hi x8 | lo x8], three
// MFMAs per product in conv_igemm's order ah*bh, ah*bl, al*bh).
// WAVES waves per workgroup, each owning 2 rows x 16 columns of the (2 * WAVES) x 16 output tile: 4 for bf16 (75 KB of
// LDS, two workgroups per CU), 8 for split-bf16 (one 159-KB workgroup per CU, two waves per SIMD).
template <bool X3, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) stem_direct_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, float *__restrict__ y, int n,
                                                          int hi, int wi, int ho, int wo, int kp, int relu) {
  constexpr int NT = 64 * WAVES, kStemTH = 2 * WAVES, kStemPR = 2 * kStemTH + 5;  // threads; tile rows; patch rows
  constexpr int GB = X3 ? 32 : 16;                    // bytes per 8-element group
  // weight row stride in LDS: 28 groups + padding so that the rows of a 16-lane ds_read_b128 group fall on
  // distinct 4-bank slots (stride in dwords = 4 mod 64)
  constexpr int WROW = X3 ? 1040 : 528;
  constexpr int OPX = X3 ? 256 : 128;                 // output bytes per pixel (64 channels)
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * WROW + kStemPR * kStemPC * GB + 32 * WAVES * 68 * 4];
  unsigned char *Ws = smem;
  unsigned char *Ps = smem + 64 * WROW;
  float *Cs = reinterpret_cast<float *>(smem + 64 * WROW + kStemPR * kStemPC * GB);  // [32 * WAVES][68] fp32 staging
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wpairs = (wi + 1) >> 1;

  // weights -> LDS once per workgroup (64 rows x 28 groups)
  constexpr int WCH = 28 * GB / 16;  // 16-byte chunks per row
  for (int c = tid; c < 64 * WCH; c += NT) {
    const int row = c / WCH, ch = c - row * WCH;
    *reinterpret_cast<u32x4 *>(Ws + row * WROW + ch * 16) = *reinterpret_cast<const u32x4 *>(
        reinterpret_cast<const unsigned char *>(w) + (size_t)row * kp * (GB / 8) + ch * 16);
  }
  const unsigned x_frame = (unsigned)hi * wpairs * GB, y_frame = (unsigned)ho * wo * OPX;  // bytes per frame
  const float floor_ = relu ? 0.f : -INFINITY;

  const int tiles_x = (wo + kStemTW - 1) / kStemTW, tiles_y = (ho + kStemTH - 1) / kStemTH;
  const long n_tiles = (long)n * tiles_y * tiles_x;
  // this lane's pixel inside the wave's 2 x 16 slice of the tile, and its A-fragment base inside the patch
  const int pr = 2 * wave + (l31 >> 4), pc = l31 & 15;
  const unsigned char *a_base = Ps + ((2 * pr) * kStemPC + pc) * GB;
  const unsigned char *b_base = Ws + l31 * WROW;

  // The patch of tile t+1 is fetched into registers while tile t is multiplied and stored (its global-load
  // latency would otherwise be exposed once per tile: there is no K loop to hide it under).
  constexpr int PCH = kStemPR * kStemPC * GB / 16;   // 16-byte chunks of a patch
  constexpr int PPASS = (PCH + NT - 1) / NT;
  u32x4 pre[PPASS];
  auto fetch_patch = [&](long t) {
    const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), f = (int)(t / ((long)tiles_x * tiles_y));
    const int iy0 = 2 * ty * kStemTH - 3, pc0 = tx * kStemTW - 2;
    // descriptor rebased to the tile's frame: 32-bit offsets suffice whatever the batch
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(x) + (size_t)f * x_frame), 0, (int)x_frame, 0x00020000);
#pragma unroll
    for (int q = 0; q < PPASS; ++q) {
      const int ci = tid + q * NT;               // chunk index inside the patch
      const int g = X3 ? ci >> 1 : ci;           // group index
      const int r = g / kStemPC, c = g - r * kStemPC;
      const int iy = iy0 + r, pcx = pc0 + c;
      const bool ok = ci < PCH && (unsigned)iy < (unsigned)hi && (unsigned)pcx < (unsigned)wpairs;
      const unsigned off = (unsigned)((iy * wpairs + pcx) * GB + (X3 ? (ci & 1) * 16 : 0));
      pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)(ok ? off : kInvalid), 0, 0);
    }
  };
  if ((long)blockIdx.x < n_tiles) fetch_patch(blockIdx.x);
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), f = (int)(t / ((long)tiles_x * tiles_y));
    const int oy0 = ty * kStemTH, ox0 = tx * kStemTW;
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(y) + (size_t)f * y_frame, 0, (int)y_frame, 0x00020000);
    __syncthreads();  // previous tile's patch and staging are free (and the weights are in place)
#pragma unroll
    for (int q = 0; q < PPASS; ++q)
      if (tid + q * NT < PCH) *reinterpret_cast<u32x4 *>(Ps + (tid + q * NT) * 16) = pre[q];
    if (t + gridDim.x < n_tiles) fetch_patch(t + gridDim.x);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int s16 = 0; s16 < 14; ++s16) {
      const int g = 2 * s16 + half;           // 8-element K group: (ky, pair j) = (g / 4, g % 4)
      const unsigned char *ap = a_base + ((g >> 2) * kStemPC + (g & 3)) * GB;
      const unsigned char *bp = b_base + g * GB;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap));
      const bf16x8 bh0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp));
      const bf16x8 bh1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp + 32 * WROW));
      if constexpr (X3) {
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap + 16));
        const bf16x8 bl0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp + 16));
        const bf16x8 bl1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp + 32 * WROW + 16));
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc[1], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc[1], 0, 0, 0);
      }
    }
    // C/D layout: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * half  ->  staging [pixel][channel]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * 68 + j * 32 + l31] = acc[j][e];
    __syncthreads();
    // 32 * WAVES pixels x 64 channels: thread -> (pixel tid / 2, 32 channels = 4 groups of 8)
    {
      const int px = tid >> 1, c0 = (tid & 1) * 32;
      const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
      const bool ok = oy < ho && ox < wo;
      const unsigned base = ok ? (unsigned)((oy * wo + ox) * OPX + (c0 / 8) * GB) : kInvalid;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(Cs + px * 68 + c0 + q * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(Cs + px * 68 + c0 + q * 8 + 4);
        const f32x4 bb0 = *reinterpret_cast<const f32x4 *>(bias + c0 + q * 8);
        const f32x4 bb1 = *reinterpret_cast<const f32x4 *>(bias + c0 + q * 8 + 4);
        const float v[8] = {fmaxf(v0[0] + bb0[0], floor_), fmaxf(v0[1] + bb0[1], floor_), fmaxf(v0[2] + bb0[2], floor_),
                            fmaxf(v0[3] + bb0[3], floor_), fmaxf(v1[0] + bb1[0], floor_), fmaxf(v1[1] + bb1[1], floor_),
                            fmaxf(v1[2] + bb1[2], floor_), fmaxf(v1[3] + bb1[3], floor_)};
        if constexpr (X3) {
          u32x4 oh, ol;
#pragma unroll
          for (int wd = 0; wd < 4; ++wd) {
            unsigned hw, lw;
            split_pair(v[2 * wd], v[2 * wd + 1], &hw, &lw);
            oh[wd] = hw;
            ol[wd] = lw;
          }
          __builtin_amdgcn_raw_buffer_store_b128(oh, rsrcY, (int)(ok ? base + q * 32 : kInvalid), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(ol, rsrcY, (int)(ok ? base + q * 32 + 16 : kInvalid), 0, 0);
        } else {
          u32x4 o;
#pragma unroll
          for (int wd = 0; wd < 4; ++wd) o[wd] = pack_bf16(v[2 * wd], v[2 * wd + 1]);
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)(ok ? base + q * 16 : kInvalid), 0, 0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// stem_pool: stem_direct with the 3x3 stride-2 max-pool fused behind it (bf16 formats).  A workgroup of 8 waves owns
// a 7 x 8 tile of POOLED pixels = the 15 x 17 conv outputs under it (255 of its 256 MFMA rows; 14 % more conv work
// than the 224 an un-pooled tiling would spend) and the 35 x 20 pixel-pair input patch under those.  The conv tile
// is staged in LDS as fp32 after bias / ReLU and the format's rounding (so that the maximum is taken over exactly the
// values the separate max-pool kernel would read back), conv pixels outside the image are -inf, and 448 threads
// reduce one 8-channel group of one pooled pixel each.  The stem's 112 x 112 x 64 output (the largest tensor of the
// network) is never written or re-read.  Bit-identical to stem_direct + maxpool3x3s2.
// ---------------------------------------------------------------------------------------------
constexpr int kPoolPH = 7, kPoolPW = 8;                              // pooled tile
constexpr int kPoolCR = 2 * kPoolPH + 1, kPoolCC = 2 * kPoolPW + 1;  // conv tile 15 x 17
constexpr int kPoolPR = 2 * kPoolCR + 5, kPoolPC = kPoolCC + 3;      // input patch 35 rows x 20 pixel pairs

template <bool X3>
// (bf16: 77 760 B of LDS and <= 128 registers, so that TWO workgroups share a CU and overlap each other's phases)
__global__ void __launch_bounds__(512, X3 ? 1 : 2) stem_pool_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ y, int n,
                                                        int hi, int wi, int ho, int wo, int hp, int wp, int kp, int relu) {
  constexpr int NT = 512;
  constexpr int GB = X3 ? 32 : 16;
  constexpr int WROW = X3 ? 1040 : 464;   // weight row stride: data + padding, conflict-free ds_read_b128 over 16 rows
  constexpr int CSB = X3 ? 256 * 68 * 4 : 256 * 72 * 2;   // the conv tile: [256][68] fp32, or (bf16) [256][72] bf16 bit patterns
  constexpr int OPX = X3 ? 256 : 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * WROW + kPoolPR * kPoolPC * GB + CSB];
  unsigned char *Ws = smem;
  unsigned char *Ps = smem + 64 * WROW;
  float *Cs = reinterpret_cast<float *>(smem + 64 * WROW + kPoolPR * kPoolPC * GB);  // X3: [256][68] fp32
  unsigned short *Cs16 = reinterpret_cast<unsigned short *>(Cs);                     // bf16: [256][72] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wpairs = (wi + 1) >> 1;

  constexpr int WCH = 28 * GB / 16;
  for (int c = tid; c < 64 * WCH; c += NT) {
    const int row = c / WCH, ch = c - row * WCH;
    *reinterpret_cast<u32x4 *>(Ws + row * WROW + ch * 16) = *reinterpret_cast<const u32x4 *>(
        reinterpret_cast<const unsigned char *>(w) + (size_t)row * kp * (GB / 8) + ch * 16);
  }
  const unsigned x_frame = (unsigned)hi * wpairs * GB, y_frame = (unsigned)hp * wp * OPX;
  const float floor_ = relu ? 0.f : -INFINITY;
  const int tiles_x = (wp + kPoolPW - 1) / kPoolPW, tiles_y = (hp + kPoolPH - 1) / kPoolPH;
  const int tiles_f = tiles_x * tiles_y;
  const int n_tiles = n * tiles_f;                 // (< 2^31: checked by the launcher)

  // MFMA row i of the workgroup = conv pixel (i / 17, i % 17) of the tile; row 255 repeats the last pixel
  const int mi = wave * 32 + l31;
  const int mr = (mi < kPoolCR * kPoolCC ? mi : kPoolCR * kPoolCC - 1) / kPoolCC;
  const int mc = (mi < kPoolCR * kPoolCC ? mi : kPoolCR * kPoolCC - 1) - mr * kPoolCC;
  const unsigned char *a_base = Ps + ((2 * mr) * kPoolPC + mc) * GB;
  const unsigned char *b_base = Ws + l31 * WROW;

  constexpr int PCH = kPoolPR * kPoolPC * GB / 16;
  constexpr int PPASS = (PCH + NT - 1) / NT;
  u32x4 pre[PPASS];
  auto fetch_patch = [&](int t) {
    const int f = t / tiles_f, rem = t - f * tiles_f, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    // conv tile origin (2*py0 - 1, 2*px0 - 1)  ->  input rows from 2*(2*py0 - 1) - 3, pairs from (2*px0 - 1) - 2
    const int iy0 = 4 * ty * kPoolPH - 5, pc0 = 2 * tx * kPoolPW - 3;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(x) + (size_t)f * x_frame), 0, (int)x_frame, 0x00020000);
#pragma unroll
    for (int q = 0; q < PPASS; ++q) {
      const int ci = tid + q * NT;
      const int g = X3 ? ci >> 1 : ci;
      const int r = g / kPoolPC, c = g - r * kPoolPC;
      const int iy = iy0 + r, pcx = pc0 + c;
      const bool ok = ci < PCH && (unsigned)iy < (unsigned)hi && (unsigned)pcx < (unsigned)wpairs;
      const unsigned off = (unsigned)((iy * wpairs + pcx) * GB + (X3 ? (ci & 1) * 16 : 0));
      pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)(ok ? off : kInvalid), 0, 0);
    }
  };
  if ((int)blockIdx.x < n_tiles) fetch_patch((int)blockIdx.x);
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int f = t / tiles_f, rem = t - f * tiles_f, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int py0 = ty * kPoolPH, px0 = tx * kPoolPW;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;   // conv pixel of tile position (0, 0)
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(y) + (size_t)f * y_frame, 0, (int)y_frame, 0x00020000);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PPASS; ++q)
      if (tid + q * NT < PCH) *reinterpret_cast<u32x4 *>(Ps + (tid + q * NT) * 16) = pre[q];
    if (t + gridDim.x < n_tiles) fetch_patch(t + gridDim.x);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int s16 = 0; s16 < 14; ++s16) {
      const int g = 2 * s16 + half;
      const unsigned char *ap = a_base + ((g >> 2) * kPoolPC + (g & 3)) * GB;
      const unsigned char *bp = b_base + g * GB;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap));
      const bf16x8 bh0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp));
      const bf16x8 bh1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp + 32 * WROW));
      if constexpr (X3) {
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap + 16));
        const bf16x8 bl0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp + 16));
        const bf16x8 bl1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp + 32 * WROW + 16));
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl0, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl1, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh1, acc[1], 0, 0, 0);
      } else {
        // bf16: the product TRANSPOSED (A = weights, B = pixels; the same products in the same order per accumulator):
        // a lane then owns ONE conv pixel and 4-channel groups, which makes the epilogue below cheap (the stem is bound
        // by its vector-ALU instruction count: 376 per wave and tile against 28 MFMAs before this, PMC)
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, ah, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, ah, acc[1], 0, 0, 0);
      }
    }
    // conv tile -> LDS as the values the format would hold: bias, ReLU, round (bf16) or split + re-sum (split-bf16);
    // conv pixels outside the image become -inf so that they never win the maximum (max-pool padding)
    if constexpr (X3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float bcol = bias[j * 32 + l31];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          const int r = row / kPoolCC, c = row - r * kPoolCC;
          const bool inside = row < kPoolCR * kPoolCC && (unsigned)(oy0 + r) < (unsigned)ho && (unsigned)(ox0 + c) < (unsigned)wo;
          float v = fmaxf(acc[j][e] + bcol, floor_);
          unsigned hw, lw;
          split_pair(v, 0.f, &hw, &lw);
          v = __builtin_bit_cast(float, hw << 16) + __builtin_bit_cast(float, lw << 16);
          Cs[row * 68 + j * 32 + l31] = inside ? v : -INFINITY;
        }
      }
    } else {
      // lane = conv pixel mi (one inside-test), acc[j][4 q + i] = channel 32 j + 8 q + 4 half + i: four bf16 bit patterns
      // (0xFF80 = -inf) per 8-byte LDS write
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const bool inside = mi < kPoolCR * kPoolCC && (unsigned)(oy0 + mr) < (unsigned)ho && (unsigned)(ox0 + mc) < (unsigned)wo;
      const u32x2 ninf = {0xFF80FF80u, 0xFF80FF80u};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + j * 32 + 8 * q + 4 * half);
          u32x2 o;
          o[0] = pack_bf16(fmaxf(acc[j][4 * q] + b[0], floor_), fmaxf(acc[j][4 * q + 1] + b[1], floor_));
          o[1] = pack_bf16(fmaxf(acc[j][4 * q + 2] + b[2], floor_), fmaxf(acc[j][4 * q + 3] + b[3], floor_));
          *reinterpret_cast<u32x2 *>(Cs16 + mi * 72 + j * 32 + 8 * q + 4 * half) = inside ? o : ninf;
        }
    }
    __syncthreads();
    if (tid < kPoolPH * kPoolPW * 8) {  // one 8-channel group of one pooled pixel per thread
      const int pp = tid >> 3, cg = tid & 7;
      const int pyl = pp / kPoolPW, pxl = pp - pyl * kPoolPW;
      float m[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          f32x4 v0, v1;
          if constexpr (X3) {
            const float *src = Cs + ((2 * pyl + ky) * kPoolCC + 2 * pxl + kx) * 68 + cg * 8;
            v0 = *reinterpret_cast<const f32x4 *>(src);
            v1 = *reinterpret_cast<const f32x4 *>(src + 4);
          } else {
            const u32x4 pkd = *reinterpret_cast<const u32x4 *>(Cs16 + ((2 * pyl + ky) * kPoolCC + 2 * pxl + kx) * 72 + cg * 8);
            v0 = f32x4{__builtin_bit_cast(float, pkd[0] << 16), __builtin_bit_cast(float, pkd[0] & 0xFFFF0000u),
                       __builtin_bit_cast(float, pkd[1] << 16), __builtin_bit_cast(float, pkd[1] & 0xFFFF0000u)};
            v1 = f32x4{__builtin_bit_cast(float, pkd[2] << 16), __builtin_bit_cast(float, pkd[2] & 0xFFFF0000u),
                       __builtin_bit_cast(float, pkd[3] << 16), __builtin_bit_cast(float, pkd[3] & 0xFFFF0000u)};
          }
          m[0] = fmaxf(m[0], v0[0]); m[1] = fmaxf(m[1], v0[1]); m[2] = fmaxf(m[2], v0[2]); m[3] = fmaxf(m[3], v0[3]);
          m[4] = fmaxf(m[4], v1[0]); m[5] = fmaxf(m[5], v1[1]); m[6] = fmaxf(m[6], v1[2]); m[7] = fmaxf(m[7], v1[3]);
        }
      const int py = py0 + pyl, px = px0 + pxl;
      const bool ok = py < hp && px < wp;
      const unsigned base = ok ? (unsigned)((py * wp + px) * OPX + cg * GB) : kInvalid;
      if constexpr (X3) {
        u32x4 oh, ol;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
          unsigned hw, lw;
          split_pair(m[2 * wd], m[2 * wd + 1], &hw, &lw);
          oh[wd] = hw;
          ol[wd] = lw;
        }
        __builtin_amdgcn_raw_buffer_store_b128(oh, rsrcY, (int)base, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(ol, rsrcY, (int)(ok ? base + 16 : kInvalid), 0, 0);
      } else {
        u32x4 o;
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) o[wd] = pack_bf16(m[2 * wd], m[2 * wd + 1]);
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)base, 0, 0);
      }
    }
  }
}

// fp32 form of stem_pool.  Input NHWC4 (one 16-byte group per pixel), weights [64][Kp] fp32 with K = (ky, kx, c4);
// the conv tile's patch is 35 rows x 39 pixels.  MFMA sequence = conv_igemm's fp32 stem exactly: v_mfma_f32_32x32x2_f32
// sums k = {4 * tap + s of lane-half 0, of lane-half 1}, taps taken in pairs (2g, 2g + 1), s = 0..3, g = 0..24, so
// the results are bit-identical to it (tap 49 is K padding: zero weights, its A operand re-reads tap 48).
constexpr int kPoolPCF = 2 * (kPoolCC - 1) + 7;   // 39 input pixels per patch row

__global__ void __launch_bounds__(512) stem_pool_f32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ y, int n,
                                                            int hi, int wi, int ho, int wo, int hp, int wp, int kp,
                                                            int relu) {
  constexpr int NT = 512;
  constexpr int WROW = 1040;                      // 200 used floats + padding: row stride = 4 dwords mod 64
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * WROW + kPoolPR * kPoolPCF * 16 + 256 * 68 * 4];
  unsigned char *Ws = smem;
  unsigned char *Ps = smem + 64 * WROW;
  float *Cs = reinterpret_cast<float *>(smem + 64 * WROW + kPoolPR * kPoolPCF * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;

  for (int c = tid; c < 64 * 50; c += NT) {       // 50 chunks of 16 B = taps 0..49 (tap 49 = zero padding)
    const int row = c / 50, ch = c - row * 50;
    *reinterpret_cast<f32x4 *>(Ws + row * WROW + ch * 16) = *reinterpret_cast<const f32x4 *>(w + (size_t)row * kp + ch * 4);
  }
  const unsigned x_frame = (unsigned)hi * wi * 16, y_frame = (unsigned)hp * wp * 256;
  const float floor_ = relu ? 0.f : -INFINITY;
  const int tiles_x = (wp + kPoolPW - 1) / kPoolPW, tiles_y = (hp + kPoolPH - 1) / kPoolPH;
  const long n_tiles = (long)n * tiles_y * tiles_x;

  const int mi = wave * 32 + l31;
  const int mr = (mi < kPoolCR * kPoolCC ? mi : kPoolCR * kPoolCC - 1) / kPoolCC;
  const int mc = (mi < kPoolCR * kPoolCC ? mi : kPoolCR * kPoolCC - 1) - mr * kPoolCC;
  const unsigned char *a_base = Ps + ((2 * mr) * kPoolPCF + 2 * mc) * 16;
  const unsigned char *b_base = Ws + l31 * WROW;

  constexpr int PCH = kPoolPR * kPoolPCF;
  constexpr int PPASS = (PCH + NT - 1) / NT;
  u32x4 pre[PPASS];
  auto fetch_patch = [&](long t) {
    const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), f = (int)(t / ((long)tiles_x * tiles_y));
    const int iy0 = 4 * ty * kPoolPH - 5, ix0 = 4 * tx * kPoolPW - 5;   // 2 * (2 * p0 - 1) - 3
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(x) + (size_t)f * x_frame), 0, (int)x_frame, 0x00020000);
#pragma unroll
    for (int q = 0; q < PPASS; ++q) {
      const int ci = tid + q * NT;
      const int r = ci / kPoolPCF, c = ci - r * kPoolPCF;
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = ci < PCH && (unsigned)iy < (unsigned)hi && (unsigned)ix < (unsigned)wi;
      pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)(ok ? (unsigned)((iy * wi + ix) * 16) : kInvalid), 0, 0);
    }
  };
  if ((long)blockIdx.x < n_tiles) fetch_patch(blockIdx.x);
  for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), f = (int)(t / ((long)tiles_x * tiles_y));
    const int py0 = ty * kPoolPH, px0 = tx * kPoolPW;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(y) + (size_t)f * y_frame, 0, (int)y_frame, 0x00020000);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PPASS; ++q)
      if (tid + q * NT < PCH) *reinterpret_cast<u32x4 *>(Ps + (tid + q * NT) * 16) = pre[q];
    if (t + gridDim.x < n_tiles) fetch_patch(t + gridDim.x);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int g = 0; g < 25; ++g) {
      const int tap = 2 * g + half;                       // 0..49; 49 is K padding (zero weights)
      const int tapa = tap < 49 ? tap : 48;               // its A operand must still be a finite number
      const int ky = tapa / 7, kx = tapa - ky * 7;
      const f32x4 a = *reinterpret_cast<const f32x4 *>(a_base + (ky * kPoolPCF + kx) * 16);
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b_base + tap * 16);
      const f32x4 b1 = *reinterpret_cast<const f32x4 *>(b_base + 32 * WROW + tap * 16);
      // Channels 0..2 only.  The generic kernel also multiplies the packed input's fourth channel -- +0 in the input
      // (pack_input / tsm_preprocess write it) times +0 in the packed weights -- which leaves every accumulator bit as it
      // is: an fp32 accumulator that starts at +0 is never -0 under round-to-nearest (x + (-x) and (+0) + (-0) are +0),
      // so acc + (+0) == acc.  A quarter of the stem's MFMAs were those (K = 49 taps x 4 -> x 3: 200 -> 150 per tile).
#pragma unroll
      for (int s4 = 0; s4 < 3; ++s4) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], b0[s4], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], b1[s4], acc[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float bcol = bias[j * 32 + l31];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        const int r = row / kPoolCC, c = row - r * kPoolCC;
        const bool inside = row < kPoolCR * kPoolCC && (unsigned)(oy0 + r) < (unsigned)ho && (unsigned)(ox0 + c) < (unsigned)wo;
        Cs[row * 68 + j * 32 + l31] = inside ? fmaxf(acc[j][e] + bcol, floor_) : -INFINITY;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {                          // 56 pooled pixels x 16 channel quads = 896 items
      const int item = tid + q * NT;
      if (item < kPoolPH * kPoolPW * 16) {
        const int pp = item >> 4, cq = item & 15;
        const int pyl = pp / kPoolPW, pxl = pp - pyl * kPoolPW;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(Cs + ((2 * pyl + ky) * kPoolCC + 2 * pxl + kx) * 68 + cq * 4);
            m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
          }
        const int py = py0 + pyl, px = px0 + pxl;
        const bool ok = py < hp && px < wp;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m), rsrcY,
                                               (int)(ok ? (unsigned)((py * wp + px) * 256 + cq * 16) : kInvalid), 0, 0);
      }
    }
  }
}

hipError_t launch_stem_pool(const float *x, const float *w, const float *bias, float *y, int n, int hi, int wi, int kp,
                            int relu, int prec, hipStream_t s) {
  const int ho = (hi + 6 - 7) / 2 + 1, wo = (wi + 6 - 7) / 2 + 1;
  const int hp = (ho + 2 - 3) / 2 + 1, wp = (wo + 2 - 3) / 2 + 1;
  if (!x || !w || !bias || !y || n <= 0 || hi <= 0 || wi <= 0 || kp < 200) return hipErrorInvalidValue;
  if (prec != kPrecBf16 && prec != kPrecBf16x3 && prec != kPrecF32) return hipErrorInvalidValue;
  if ((double)hi * wi * 16.0 > 2.0e9 || (prec != kPrecF32 && kp < 224)) return hipErrorInvalidValue;
  const long tiles = (long)n * ((hp + kPoolPH - 1) / kPoolPH) * ((wp + kPoolPW - 1) / kPoolPW);
  if (tiles >= (1L << 31) - 1024) return hipErrorInvalidValue;
  const long cap = (long)device_info().n_cu * (prec == kPrecBf16 ? 2 : 1);   // persistent: one 8-wave workgroup per CU (bf16: two)
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
  if (prec == kPrecF32)
    hipLaunchKernelGGL(stem_pool_f32_kernel, dim3(grid), dim3(512), 0, s, x, w, bias, y, n, hi, wi, ho, wo, hp, wp, kp, relu);
  else if (prec == kPrecBf16)
    hipLaunchKernelGGL(stem_pool_kernel<false>, dim3(grid), dim3(512), 0, s, x, w, bias, y, n, hi, wi, ho, wo, hp, wp, kp, relu);
  else
    hipLaunchKernelGGL(stem_pool_kernel<true>, dim3(grid), dim3(512), 0, s, x, w, bias, y, n, hi, wi, ho, wo, hp, wp, kp, relu);
  return hipGetLastError();
}

hipError_t launch_stem_direct(const float *x, const float *w, const float *bias, float *y, int n, int hi, int wi, int kp,
                              int relu, int prec, hipStream_t s) {
  const int ho = (hi + 6 - 7) / 2 + 1, wo = (wi + 6 - 7) / 2 + 1;
  if (!x || !w || !bias || !y || n <= 0 || hi <= 0 || wi <= 0 || kp < 224) return hipErrorInvalidValue;
  if (prec != kPrecBf16 && prec != kPrecBf16x3) return hipErrorInvalidValue;
  if ((double)hi * ((wi + 1) / 2) * 32.0 > 2.0e9 || (double)ho * wo * 256.0 > 2.0e9) return hipErrorInvalidValue;
  const int th = prec == kPrecBf16 ? 8 : 16;
  const long tiles = (long)n * ((ho + th - 1) / th) * ((wo + kStemTW - 1) / kStemTW);
  // persistent workgroups: two per CU for bf16 (4 waves, 75 KB of LDS each), one per CU for split-bf16 (8 waves, 159 KB)
  const long cap = (long)device_info().n_cu * (prec == kPrecBf16 ? 2 : 1);
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
  if (prec == kPrecBf16)
    hipLaunchKernelGGL((stem_direct_kernel<false, 4>), dim3(grid), dim3(256), 0, s, x, w, bias, y, n, hi, wi, ho, wo, kp, relu);
  else
    hipLaunchKernelGGL((stem_direct_kernel<true, 8>), dim3(grid), dim3(512), 0, s, x, w, bias, y, n, hi, wi, ho, wo, kp, relu);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Split-K reduction (fp32): the segment sums of a ksplit launch are added in segment order -- the order the
// unsplit kernel uses -- then bias, residual and ReLU exactly as in the conv epilogue.  4 channels per thread.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float *__restrict__ partial, int n_seg, int64_t n4,
                                                            int64_t seg_stride4, int cout4,
                                                            const float *__restrict__ bias, const float *__restrict__ res,
                                                            float *__restrict__ y, int relu) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float floor_ = relu ? 0.f : -INFINITY;
  const f32x4 *p4 = reinterpret_cast<const f32x4 *>(partial);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 t = p4[i];
    for (int sgm = 1; sgm < n_seg; ++sgm) t += p4[i + sgm * seg_stride4];
    f32x4 v = t + reinterpret_cast<const f32x4 *>(bias)[i % cout4];
    if (res) v += reinterpret_cast<const f32x4 *>(res)[i];
    v[0] = fmaxf(v[0], floor_);
    v[1] = fmaxf(v[1], floor_);
    v[2] = fmaxf(v[2], floor_);
    v[3] = fmaxf(v[3], floor_);
    reinterpret_cast<f32x4 *>(y)[i] = v;
  }
}

static unsigned grid_for(int64_t total, int cap);

hipError_t launch_splitk_reduce(const float *partial, int n_seg, int64_t m, int cout, const float *bias,
                                const float *res, float *y, int relu, hipStream_t s) {
  if (!partial || !bias || !y || n_seg < 1 || m <= 0 || cout <= 0 || cout % 4 != 0) return hipErrorInvalidValue;
  const int64_t n4 = m * cout / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n4, 2048)), dim3(256), 0, s, partial, n_seg, n4, n4, cout / 4,
                     bias, res, y, relu);
  return hipGetLastError();
}

// =============================================================================================
// Storage formats of activations outside the conv kernel.  A "group" is the unit one thread moves:
//   kPrecF32     4 channels, 16 bytes (4 floats)
//   kPrecBf16x3  8 channels, 32 bytes [hi x8 | lo x8] (split-bf16)
//   kPrecBf16    8 channels, 16 bytes (8 bf16)
// Pointers stay float-typed; gf = group size in 4-byte units.
// =============================================================================================
template <int FMT>
struct Fmt {
  static constexpr int ch = FMT == kPrecF32 ? 4 : 8;
  static constexpr int gf = FMT == kPrecBf16x3 ? 8 : 4;
};

template <int FMT>
__device__ __forceinline__ void load_group(const float *p, float v[8]) {
  if (FMT == kPrecF32) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = a[e];
      v[e + 4] = 0.f;
    }
  } else if (FMT == kPrecBf16x3) {
    const u32x4 h = *reinterpret_cast<const u32x4 *>(p), l = *reinterpret_cast<const u32x4 *>(p + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = split_elem(h, e) + split_elem(l, e);
  } else {
    const u32x4 h = *reinterpret_cast<const u32x4 *>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = split_elem(h, e);
  }
}

template <int FMT>
__device__ __forceinline__ void store_group(float *p, const float v[8]) {
  if (FMT == kPrecF32) {
    *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
  } else if (FMT == kPrecBf16x3) {
    u32x4 oh, ol;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      unsigned hw, lw;
      split_pair(v[2 * w], v[2 * w + 1], &hw, &lw);
      oh[w] = hw;
      ol[w] = lw;
    }
    *reinterpret_cast<u32x4 *>(p) = oh;
    *reinterpret_cast<u32x4 *>(p + 4) = ol;
  } else {
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) o[w] = pack_bf16(v[2 * w], v[2 * w + 1]);
    *reinterpret_cast<u32x4 *>(p) = o;
  }
}

static unsigned grid_for(int64_t total, int cap) {
  const int64_t blocks = (total + 255) / 256;
  return (unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

#define TSM_DISPATCH_FMT(prec, KERNEL, grid, stream, ...)                                                   \
  do {                                                                                                      \
    if ((prec) == kPrecBf16x3)                                                                              \
      hipLaunchKernelGGL((KERNEL<kPrecBf16x3>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);             \
    else if ((prec) == kPrecBf16)                                                                           \
      hipLaunchKernelGGL((KERNEL<kPrecBf16>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);               \
    else                                                                                                    \
      hipLaunchKernelGGL((KERNEL<kPrecF32>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                \
  } while (0)

// ---------------------------------------------------------------------------------------------
// pack_input: [N,3,H,W] or [N,H,W,3] fp32 -> the stem's input format, padding channels zero:
//   fp32   one 4-channel group per pixel (NHWC4)
//   bf16 formats   one 8-element group per pixel PAIR: (pixel 2j: c0 c1 c2 0, pixel 2j+1: c0 c1 c2 0), rows of
//                  ceil(W/2) pairs (an odd width ends in a zero pixel, which is what the conv's padding reads anyway)
// One thread per group.
// ---------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256) pack_input_kernel(const float *__restrict__ src,
                                                         float *__restrict__ dst, int64_t n_groups, int h, int w,
                                                         int nchw) {
  constexpr int PX = FMT == kPrecF32 ? 1 : 2;  // pixels per group
  const int wg = (w + PX - 1) / PX;
  const int64_t hw = (int64_t)h * w;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_groups; i += stride) {
    const int gx = (int)(i % wg);
    const int64_t row = i / wg;  // n * h + y
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < PX; ++q) {
      const int x = gx * PX + q;
      if (x < w) {
        if (nchw) {
          const int64_t n = row / h, y = row - n * h;
          const float *b = src + n * 3 * hw + y * w + x;
          v[4 * q + 0] = b[0];
          v[4 * q + 1] = b[hw];
          v[4 * q + 2] = b[2 * hw];
        } else {
          const float *b = src + (row * w + x) * 3;
          v[4 * q + 0] = b[0];
          v[4 * q + 1] = b[1];
          v[4 * q + 2] = b[2];
        }
      }
    }
    store_group<FMT>(dst + i * Fmt<FMT>::gf, v);
  }
}

hipError_t launch_pack_input(const float *src, float *dst, int64_t n_frames, int h, int w, int nchw, int prec,
                             hipStream_t s) {
  const int wg = prec == kPrecF32 ? w : (w + 1) / 2;
  const int64_t total = n_frames * h * wg;
  TSM_DISPATCH_FMT(prec, pack_input_kernel, grid_for(total, 4096), s, src, dst, total, h, w, nchw);
  return hipGetLastError();
}

// fp32 [n8 * 8] <-> another format, 8 channels per thread
template <int FMT>
__global__ void __launch_bounds__(256) from_f32_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                       int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(x + i * 8), b = *reinterpret_cast<const f32x4 *>(x + i * 8 + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    store_group<FMT>(y + i * Fmt<FMT>::gf, v);
  }
}
template <int FMT>
__global__ void __launch_bounds__(256) to_f32_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                     int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float v[8];
    load_group<FMT>(x + i * Fmt<FMT>::gf, v);
    *reinterpret_cast<f32x4 *>(y + i * 8) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4 *>(y + i * 8 + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}
hipError_t launch_from_f32(const float *x, float *y, int64_t n8, int prec, hipStream_t s) {
  if (prec == kPrecBf16x3)
    hipLaunchKernelGGL(from_f32_kernel<kPrecBf16x3>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else if (prec == kPrecBf16)
    hipLaunchKernelGGL(from_f32_kernel<kPrecBf16>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_to_f32(const float *x, float *y, int64_t n8, int prec, hipStream_t s) {
  if (prec == kPrecBf16x3)
    hipLaunchKernelGGL(to_f32_kernel<kPrecBf16x3>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else if (prec == kPrecBf16)
    hipLaunchKernelGGL(to_f32_kernel<kPrecBf16>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// preprocess (K8): one thread per output pixel.  Bilinear sampling follows ATen's CPU kernel
// (UpSampleBilinear2d): src = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out,
// out = h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11); then (v*pre_scale - mean)/std.
// datasets/build.py:131-136 of the reference (torchvision tensor transforms).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void preprocess_pixel(const PreprocParams &p, const T *frame, int cy, int cx, float *v) {
  const float sh = (float)p.h / (float)p.nh, sw = (float)p.w / (float)p.nw;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  float fy = sh * ((float)(cy + p.top) + 0.5f) - 0.5f;
  float fx = sw * ((float)(cx + p.left) + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
  const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
  const T *r0 = frame + (int64_t)y0 * p.w * 3, *r1 = frame + (int64_t)y1 * p.w * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = (float)r0[x0 * 3 + c], p01 = (float)r0[x1 * 3 + c];
    const float p10 = (float)r1[x0 * 3 + c], p11 = (float)r1[x1 * 3 + c];
    const float t = h0 * (w0 * p00 + w1 * p01) + h1 * (w0 * p10 + w1 * p11);
    v[c] = (t * p.pre_scale - mean[c]) / stdv[c];
  }
}

// One thread per output group: a pixel (out_mode 0 NHWC4 fp32, 1 NCHW fp32) or a pixel pair (2 split-bf16,
// 3 bf16: the stem's packed-pair input, see pack_input_kernel).
template <typename T>
__global__ void __launch_bounds__(256) preprocess_kernel(const PreprocParams p) {
  const int px = p.out_mode >= 2 ? 2 : 1;
  const int wg = (p.crop + px - 1) / px;
  const int64_t total = (int64_t)p.n * p.crop * wg;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const T *src = static_cast<const T *>(p.src);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int gx = (int)(i % wg);
    const int cy = (int)((i / wg) % p.crop);
    const int64_t f = i / ((int64_t)wg * p.crop);
    const T *frame = src + f * (int64_t)p.h * p.w * 3;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    preprocess_pixel<T>(p, frame, cy, gx * px, v);
    if (px == 2 && gx * 2 + 1 < p.crop) preprocess_pixel<T>(p, frame, cy, gx * 2 + 1, v + 4);
    if (p.out_mode == 1) {
      float *o = p.dst + f * 3 * (int64_t)p.crop * p.crop + (int64_t)cy * p.crop + gx;
      o[0] = v[0];
      o[(int64_t)p.crop * p.crop] = v[1];
      o[2 * (int64_t)p.crop * p.crop] = v[2];
    } else if (p.out_mode == 2) {
      store_group<kPrecBf16x3>(p.dst + i * 8, v);
    } else if (p.out_mode == 3) {
      store_group<kPrecBf16>(p.dst + i * 4, v);
    } else {
      store_group<kPrecF32>(p.dst + i * 4, v);
    }
  }
}

hipError_t launch_preprocess(const PreprocParams &p, hipStream_t s) {
  if (p.n <= 0 || p.h <= 0 || p.w <= 0 || p.crop <= 0 || p.top < 0 || p.left < 0 || p.top + p.crop > p.nh ||
      p.left + p.crop > p.nw)
    return hipErrorInvalidValue;
  const int px = p.out_mode >= 2 ? 2 : 1;
  const int64_t total = (int64_t)p.n * p.crop * ((p.crop + px - 1) / px);
  const unsigned grid = grid_for(total, 8192);
  if (p.src_is_u8)
    hipLaunchKernelGGL(preprocess_kernel<unsigned char>, dim3(grid), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3(grid), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gather_clips: the clip iterator of the dataset loop (utils/inference_count.py:411-414, video[i:i + 16:2] with a
// zero-padded tail) over TRANSFORMED frames that sit in a device buffer: out[c][k] = frame of source index
// step * (first_clip + c) + stride * k, the buffer's pad frame where that index is past the video's end.  Frames are
// opaque rows of frame_bytes (any packed layout); 16-byte copies, four in flight per thread.  HBM-bound and tiny
// next to the forward (a batch of 32 clips moves 2 x 205 MB: 0.1 ms) -- it exists so that the loop's only device work
// between two forwards is this library's: torch's index_select costs two first-use code-object loads (4 + 150 ms with
// the GPU idle at the head of every cold dataset job, profiles/r03_config4_gpu_gaps_pieces.txt).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_clips_kernel(const GatherParams p) {
  const int64_t row = p.row0 + blockIdx.y;          // (clip, segment) of the output
  const int64_t c = row / p.n_segment;
  const int k = (int)(row - c * p.n_segment);
  const int64_t src_frame = (int64_t)p.clip_step * (p.first_clip + c) + (int64_t)p.clip_stride * k;
  const int64_t j = src_frame < p.total_frames ? src_frame / p.clip_stride - p.first_frame : p.pad_frame;
  const uint4 *src = reinterpret_cast<const uint4 *>(static_cast<const char *>(p.frames) + j * p.frame_bytes);
  uint4 *dst = reinterpret_cast<uint4 *>(static_cast<char *>(p.out) + row * p.frame_bytes);
  const int64_t n16 = p.frame_bytes / 16;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n16) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n16) dst[i + u * 256] = v[u];
  }
}

hipError_t launch_gather_clips(const GatherParams &p_in, hipStream_t s) {
  GatherParams p = p_in;
  if (!p.frames || !p.out || p.n_frames <= 0 || p.frame_bytes <= 0 || p.frame_bytes % 16 != 0 || p.n_clips <= 0 ||
      p.n_segment <= 0 || p.clip_step <= 0 || p.clip_stride <= 0 || p.clip_step % p.clip_stride != 0 ||
      p.first_clip < 0 || p.first_frame < 0 || p.total_frames <= 0)
    return hipErrorInvalidValue;
  // every index the kernel will form, checked here: the first and the last in-video position of the range, and the pad frame
  const int64_t lo = (int64_t)p.clip_step * p.first_clip;
  const int64_t hi = (int64_t)p.clip_step * (p.first_clip + p.n_clips - 1) + (int64_t)p.clip_stride * (p.n_segment - 1);
  if (lo >= p.total_frames) return hipErrorInvalidValue;                       // a clip starts inside its video
  const int64_t last = (hi < p.total_frames ? hi : p.total_frames - 1) / p.clip_stride - p.first_frame;
  const int64_t first = lo / p.clip_stride - p.first_frame;
  if (first < 0 || last >= p.n_frames) return hipErrorInvalidValue;
  // a padded tail reads the pad frame: it must lie in the buffer and must not be one of the range's own video frames (a
  // mis-sized buffer or a wrong first_frame / total_frames pair would otherwise pass a real frame off as the zero frame)
  if (hi >= p.total_frames && (p.pad_frame < 0 || p.pad_frame >= p.n_frames || (p.pad_frame >= first && p.pad_frame <= last)))
    return hipErrorInvalidValue;
  const int64_t n16 = p.frame_bytes / 16;
  const unsigned gx = (unsigned)((n16 + 1023) / 1024 < 64 ? (n16 + 1023) / 1024 : 64);
  // grid.y holds at most 65535 (clip, segment) rows: longer ranges are cut into several launches here, so that the limit
  // is not a property of the C ABI
  const int64_t rows = (int64_t)p.n_clips * p.n_segment;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    p.row0 = r0;
    const int64_t ny = rows - r0 < 65535 ? rows - r0 : 65535;
    hipLaunchKernelGGL(gather_clips_kernel, dim3(gx, (unsigned)ny), dim3(256), 0, s, p);
    const hipError_t st = hipGetLastError();
    if (st != hipSuccess) return st;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// maxpool 3x3 stride 2 pad 1, NHWC; one thread per (output pixel, channel group).
// ---------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float *__restrict__ x,
                                                           float *__restrict__ y, int n, int hi, int wi,
                                                           int ho, int wo, int cg) {
  constexpr int GF = Fmt<FMT>::gf;
  const int64_t total = (int64_t)n * ho * wo * cg;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int g = (int)(i % cg);
    int64_t pix = i / cg;
    const int ox = (int)(pix % wo);
    pix /= wo;
    const int oy = (int)(pix % ho);
    const int64_t f = pix / ho;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)hi) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)wi) continue;
        float v[8];
        load_group<FMT>(x + (((f * hi + iy) * wi + ix) * cg + g) * GF, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    store_group<FMT>(y + i * GF, m);
  }
}

hipError_t launch_maxpool3x3s2(const float *x, float *y, int n, int hi, int wi, int c, int prec, hipStream_t s) {
  const int gch = prec == kPrecF32 ? 4 : 8;
  if (c % gch != 0) return hipErrorInvalidValue;
  const int ho = (hi + 2 - 3) / 2 + 1, wo = (wi + 2 - 3) / 2 + 1;
  const int cg = c / gch;
  const int64_t total = (int64_t)n * ho * wo * cg;
  TSM_DISPATCH_FMT(prec, maxpool3x3s2_kernel, grid_for(total, 8192), s, x, y, n, hi, wi, ho, wo, cg);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Stand-alone temporal shift (NHWC fp32).  One thread per 16-B channel quad; fold % 4 == 0 so a quad
// never straddles a fold boundary.  tsm.py:35-50.  (The forward uses the conv kernel's fused loader.)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) temporal_shift_kernel(const float *__restrict__ x,
                                                             float *__restrict__ y, int64_t n_frames,
                                                             int n_segment, int64_t hw, int c4, int fold4) {
  const int64_t total = n_frames * hw * c4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t frame_quads = hw * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cq = (int)(i % c4);
    const int64_t f = i / frame_quads;
    const int t = (int)(f % n_segment);
    int dt = cq < fold4 ? 1 : (cq < 2 * fold4 ? -1 : 0);
    const bool ok = dt == 1 ? t < n_segment - 1 : (dt == -1 ? t > 0 : true);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4 *>(x + (i + dt * frame_quads) * 4);
    *reinterpret_cast<f32x4 *>(y + i * 4) = v;
  }
}

hipError_t launch_temporal_shift(const float *x, float *y, int64_t n_frames, int n_segment, int64_t hw,
                                 int c, int fold, hipStream_t s) {
  if (c % 4 != 0 || fold % 4 != 0 || n_segment <= 0 || n_frames % n_segment != 0) return hipErrorInvalidValue;
  const int64_t total = n_frames * hw * (c / 4);
  hipLaunchKernelGGL(temporal_shift_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, x, y, n_frames,
                     n_segment, hw, c / 4, fold / 4);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Head.  logits[b] = fc( mean_t mean_hw feat[b,t,hw,:] ) + bias  (avg-pool, FC and the segment mean
// are all linear, so pooling first is exact up to fp32 summation order).  tsm.py:411-419.
//   head_pool: grid (n_frames, ...): per-frame average pool into pooled[n_frames, c] (fp32);
//              one thread per channel group, rows streamed with 16-byte loads.
//   head_fc  : grid (n_clips, num_class) x 64 lanes: mean over the clip's frames, dot with one class row.
// ---------------------------------------------------------------------------------------------
template <int FMT>
__global__ void __launch_bounds__(256) head_pool_kernel(const float *__restrict__ feat,
                                                        float *__restrict__ pooled, int rows, int c) {
  constexpr int GF = Fmt<FMT>::gf, GC = Fmt<FMT>::ch;
  const int b = blockIdx.x;
  const int t = blockIdx.y * 256 + threadIdx.x;
  const int cg = c / GC;
  if (t >= cg) return;
  const float *src = feat + ((size_t)b * rows * cg + t) * GF;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // rows are added in order; unrolled by 7 (49 = 7 x 7 at 224^2) so that the loads of a group are in flight together
#pragma unroll 7
  for (int r = 0; r < rows; ++r) {
    float v[8];
    load_group<FMT>(src + (size_t)r * cg * GF, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < GC; ++e) pooled[(size_t)b * c + t * GC + e] = acc[e] / (float)rows;
}

// one 64-lane workgroup per (clip, class): every lane streams its share of the 2048 channels over the
// clip's T pooled frames (independent loads), then a wave reduction
__global__ void __launch_bounds__(64) head_fc_kernel(const float *__restrict__ pooled,
                                                     const float *__restrict__ fc_w,
                                                     const float *__restrict__ fc_b,
                                                     float *__restrict__ logits, int c, int num_class,
                                                     int n_segment) {
  const int b = blockIdx.x, cls = blockIdx.y;
  const int lane = threadIdx.x;
  const float *pv = pooled + (size_t)b * n_segment * c;  // per-frame pooled features of this clip
  const float *wv = fc_w + (size_t)cls * c;
  float s = 0.f;
  for (int k = lane * 4; k < c; k += 256) {
    f32x4 f = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int t = 0; t < n_segment; ++t) f += *reinterpret_cast<const f32x4 *>(pv + (size_t)t * c + k);
    const f32x4 w = *reinterpret_cast<const f32x4 *>(wv + k);
    s += (f[0] * w[0] + f[1] * w[1]) + (f[2] * w[2] + f[3] * w[3]);
  }
  s /= (float)n_segment;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) logits[(size_t)b * num_class + cls] = s + fc_b[cls];
}

hipError_t launch_head(const float *feat, const float *fc_w, const float *fc_b, float *pooled,
                       float *logits, int n_clips, int n_segment, int hw, int c, int num_class, int prec,
                       hipStream_t s) {
  if (n_clips <= 0 || c <= 0 || c % 8 != 0) return hipErrorInvalidValue;
  const int cg = c / (prec == kPrecF32 ? 4 : 8);
  const dim3 grid(n_clips * n_segment, (cg + 255) / 256);
  if (prec == kPrecBf16x3)
    hipLaunchKernelGGL(head_pool_kernel<kPrecBf16x3>, grid, dim3(256), 0, s, feat, pooled, hw, c);
  else if (prec == kPrecBf16)
    hipLaunchKernelGGL(head_pool_kernel<kPrecBf16>, grid, dim3(256), 0, s, feat, pooled, hw, c);
  else
    hipLaunchKernelGGL(head_pool_kernel<kPrecF32>, grid, dim3(256), 0, s, feat, pooled, hw, c);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(head_fc_kernel, dim3(n_clips, num_class), dim3(64), 0, s, pooled, fc_w, fc_b, logits, c,
                     num_class, n_segment);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K9: logits -> per-clip state on the GPU (utils/eval.py:153-164 + to_softmax, utils/visualize.py:140-150):
// optional fp32 softmax over the classes, FIRST maximum, class id if its score >= threshold else -1.  One thread per
// clip (n_clips x num_class is tiny; the point is that a streaming step copies 8 bytes per window to the host instead
// of the logits, and needs no host-side numpy pass).  The sum runs in numpy's order for rows of 8..128 elements
// (8 strided partial sums, a fixed tree, then the remainder), so probabilities match the host path up to the 1-ulp
// freedom of expf itself.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) scores_to_states_kernel(const float *__restrict__ logits, int n, int c, int softmax,
                                                              float threshold, int *__restrict__ states,
                                                              float *__restrict__ top) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float *s = logits + (size_t)i * c;
  float best = 0.f;
  int arg = 0;
  if (softmax) {
    float mx = s[0];
    for (int j = 1; j < c; ++j) mx = fmaxf(mx, s[j]);
    float sum;
    if (c >= 8 && c <= 128) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = expf(s[j] - mx);
      int j = 8;
      for (; j + 8 <= c; j += 8)
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] += expf(s[j + q] - mx);
      sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (; j < c; ++j) sum += expf(s[j] - mx);
    } else {
      sum = 0.f;
      for (int j = 0; j < c; ++j) sum += expf(s[j] - mx);
    }
    for (int j = 0; j < c; ++j) {
      const float p = expf(s[j] - mx) / sum;
      if (j == 0 || p > best) {
        best = p;
        arg = j;
      }
    }
  } else {
    best = s[0];
    for (int j = 1; j < c; ++j)
      if (s[j] > best) {
        best = s[j];
        arg = j;
      }
  }
  states[i] = best >= threshold ? arg : -1;
  if (top) top[i] = best;
}

hipError_t launch_scores_to_states(const float *logits, int n, int c, int softmax, float threshold, int *states, float *top,
                                   hipStream_t s) {
  if (!logits || !states || n <= 0 || c <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(scores_to_states_kernel, dim3((n + 63) / 64), dim3(64), 0, s, logits, n, c, softmax, threshold, states, top);
  return hipGetLastError();
}

}  // namespace tsm
