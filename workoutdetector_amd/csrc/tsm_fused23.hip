// conv23_fused_kernel: Bottleneck.conv2 + conv3 + residual in one launch (fp32 / split-bf16).
#include "tsm_device.h"

namespace tsm {

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
    if (cmid == 64) TSM_KLAUNCH((conv23_fused_kernel<64, true>), dim3(grid), dim3(256), 0, s, p);
    else TSM_KLAUNCH((conv23_fused_kernel<128, true>), dim3(grid), dim3(512), 0, s, p);
  } else {
    if (cmid == 64) TSM_KLAUNCH((conv23_fused_kernel<64, false>), dim3(grid), dim3(256), 0, s, p);
    else TSM_KLAUNCH((conv23_fused_kernel<128, false>), dim3(grid), dim3(512), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace tsm
