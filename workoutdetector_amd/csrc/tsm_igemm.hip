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
#include "tsm_device.h"

namespace tsm {

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
  const int bid = blockIdx.x;
  const bool tail_mode = SEG && p.ksplit == 2;            // whole tiles first, then (tile, segment) pieces of the last rows of tiles
  const bool in_tail = tail_mode && bid >= p.tail_from;
  const int nwg = tail_mode ? p.tail_from : (int)gridDim.x;   // (the remap covers the whole tiles; the pieces go round-robin over the XCDs)
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  if (p.reverse) tile = nwg - 1 - tile;   // walk the tiles from the far end (ConvParams::reverse)
  int seg = 0;  // SEG + ksplit = 1: the grid holds ntm * ntn tiles per K segment, segment-major
  if (SEG && p.ksplit == 1) {
    seg = tile / (p.ntm * p.ntn);
    tile -= seg * (p.ntm * p.ntn);
  }
  if (in_tail) {
    const int u = bid - p.tail_from, n_tail = p.ntm * p.ntn - p.tail_from;
    seg = u / n_tail;
    tile = p.tail_from + (u - seg * n_tail);
  }
  const bool split_wg = SEG && (p.ksplit == 1 || in_tail);   // this workgroup multiplies ONE K segment and writes raw sums
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
  if (split_wg) {
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
  const bool partial = split_wg;  // raw segment sums to y = partial[seg][M][Cout] (tail split: ypart = partial[seg][tail rows][Cout]): no bias, no ReLU
  const size_t y_bytes = ((size_t)p.M - m0) * p.Cout * EB;
  const int mt0 = in_tail ? (p.tail_from / p.ntn) * BM : 0;   // first row of the tail
  char *ybase = in_tail ? reinterpret_cast<char *>(p.ypart) + ((size_t)seg * (p.M - mt0) + (m0 - mt0)) * p.Cout * EB
                        : reinterpret_cast<char *>(p.y) + ((size_t)(partial ? seg : 0) * p.M + m0) * p.Cout * EB;
  const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
      ybase, 0, (int)(y_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : y_bytes), 0x00020000);
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
    const int ntiles = p.ntm * p.ntn;
    if (p.ksplit == 2 && (!p.ypart || p.tail_from <= 0 || p.tail_from >= ntiles || p.tail_from % p.ntn != 0)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(p.ksplit == 2 ? p.tail_from + (ntiles - p.tail_from) * conv_num_segments(p)
                                             : ntiles * (p.ksplit ? conv_num_segments(p) : 1)));
    const dim3 block(64 * WGM * WGN);
    if constexpr (KS == 1 && !SHIFT) {
      if (p.x2) {
        TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecF32, true, true>), grid, block, 0, s, p);
        return hipGetLastError();
      }
    }
    TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, false, kPrecF32, false, true>), grid, block, 0, s, p);
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
        TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecBf16x3, true>), grid, dim3(64 * WGM * WGN), 0, s, p);
      else if (p.prec == kPrecBf16)
        TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecBf16, true>), grid, dim3(64 * WGM * WGN), 0, s, p);
      else
        TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, 1, false, false, kPrecF32, true>), grid, dim3(64 * WGM * WGN), 0, s, p);
      return hipGetLastError();
    }
  }
  if (p.prec == kPrecBf16x3)
    TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecBf16x3>), grid, dim3(64 * WGM * WGN), 0, s, p);
  else if (p.prec == kPrecBf16)
    TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecBf16>), grid, dim3(64 * WGM * WGN), 0, s, p);
  else
    TSM_KLAUNCH((conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecF32>), grid, dim3(64 * WGM * WGN), 0, s, p);
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
  if ((p.ksplit && p.kseg_len <= 0) || p.ksplit < 0 || p.ksplit > 2) return hipErrorInvalidValue;
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


hipError_t launch_splitk_reduce(const float *partial, int n_seg, int64_t m, int cout, const float *bias,
                                const float *res, float *y, int relu, hipStream_t s) {
  if (!partial || !bias || !y || n_seg < 1 || m <= 0 || cout <= 0 || cout % 4 != 0) return hipErrorInvalidValue;
  const int64_t n4 = m * cout / 4;
  TSM_KLAUNCH(splitk_reduce_kernel, dim3(grid_for(n4, 2048)), dim3(256), 0, s, partial, n_seg, n4, n4, cout / 4,
                     bias, res, y, relu);
  return hipGetLastError();
}

}  // namespace tsm
