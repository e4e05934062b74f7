// Weight-stationary bf16 kernels for layer1 / layer2: conv3x3_ws[128]_kernel, conv1x1_ws[n]_kernel.
#include "tsm_device.h"

namespace tsm {

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
  int swz;             // conv3x3_ws128: which bit swaps the two 16-byte halves of a patch position (ws128_swap; chosen by the host per geometry)
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
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcYall, (int)(yoff + (unsigned)(nt * 64 + (2 * half + qq) * 16)), 0, TSM_AUX_WS);
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
                o, rsrcY, (int)(yo[mt] == kInvalid ? kInvalid : yo[mt] + (unsigned)(it * 64 + (2 * half + qq) * 16)), 0, TSM_AUX_WS);
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
//
// S2 = true: the same with stride 2 (layer2.0's conv2).  A tile of TR x TC <= 64 output pixels (ONE pair of M-tiles: a patch
// is four times its outputs) reads a (2 TR + 1) x (2 TC + 1) patch of at most 17 x 17 = 289 pixels: nine whole DMA rounds of
// 32 pixels per plane and a tenth of ONE pixel (lanes 0 and 1 only -- a whole round there would run 992 bytes into the next
// plane), 2 x 73 984 B of LDS.  Two things keep the pixel-fragment reads conflict-free although neighbouring outputs are two
// input columns apart: the patch is stored with its columns DE-INTERLEAVED (row pitch 2 TC + 1: the TC + 1 even columns,
// then the TC odd ones; tap kx = 0 / 2 reads position c / c + 1 of the even run, kx = 1 position c of the odd run), and an
// M-tile's lanes are laid out LPR = 8 / 16 / 32 lanes per tile row with, for LPR = 8, rows r and r + 4 in the same 16-lane
// group (rows 2 (2 TC + 1) x 4 positions apart: an odd multiple of 8, i.e. the other 16-byte half of the same banks).
// ---------------------------------------------------------------------------------------------
constexpr int kW8Rounds = 6;                      // DMA rounds of 32 patch pixels per plane
constexpr int kW8PatchMax = kW8Rounds * 32;       // 192 patch pixels (10 x 18 for an 8 x 16 tile, 6 x 30 for 4 x 28)
constexpr int kW8Plane = kW8PatchMax * 32;
constexpr int kW8BufBytes = 8 * kW8Plane;         // 49 152 B
constexpr int kW8LdsBytes = 2 * kW8BufBytes + 512;
constexpr int kW8AgprFrags = 56;                  // fragments kept in accumulation registers
constexpr int kS2PatchMax = 289;                  // 17 x 17 for an 8 x 8 tile (config 5: 32 x 32 outputs), 9 x 29 for 4 x 14 (28 x 28)
constexpr int kS2Rounds = 10;                     // the tenth: one pixel
constexpr int kS2Plane = kS2PatchMax * 32;        // 9 248 B
constexpr int kS2BufBytes = 8 * kS2Plane;         // 73 984 B
constexpr int kS2LdsBytes = 2 * kS2BufBytes + 512;

// Stride 2: lanes per tile row of an M-tile pair (64 lanes), and the tile geometry for Ho x Wo outputs: TC <= LPR columns,
// TR <= 64 / LPR rows, (2 TR + 1) x (2 TC + 1) <= kS2PatchMax patch pixels, fewest tiles per frame (ties: the smaller patch).
static int ws_s2_lanes_per_row(int tc) { return tc <= 8 ? 8 : tc <= 16 ? 16 : tc <= 32 ? 32 : 64; }
static bool ws_s2_tile_geometry(int Ho, int Wo, int *tr_out, int *tc_out) {
  long best_tiles = -1;
  int best_tr = 0, best_tc = 0, best_patch = 0;
  for (int tc = 1; tc <= 64; ++tc) {
    int tr = 64 / ws_s2_lanes_per_row(tc);
    if (tr > Ho) tr = Ho;
    if (tr < 1) continue;
    const int patch = (2 * tr + 1) * (2 * tc + 1);
    if (patch > kS2PatchMax) continue;
    if (2 * tr * (2 * tc + 1) + ws_s2_lanes_per_row(tc) + tc >= kS2PatchMax) continue;   // idle lanes of a row read behind it: the largest position they form stays <= kS2PatchMax - 1, inside the plane
    const long tiles = (long)((Ho + tr - 1) / tr) * ((Wo + tc - 1) / tc);
    if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && patch < best_patch)) {
      best_tiles = tiles; best_tr = tr; best_tc = tc; best_patch = patch;
    }
  }
  *tr_out = best_tr;
  *tc_out = best_tc;
  return best_tiles > 0;
}

// The pixel of lane l31 of M-tile mt (q = 32 mt + l31): its row / column in the tile (row 0x4000: an idle lane) and the patch
// position of its top-left tap.  One definition for the kernel and for the host's bank-conflict model below.
__host__ __device__ inline void ws128_lane_pixel(bool s2, int TR, int TC, int PW, int q, int *prow, int *pcol, int *pp0) {
  if (s2) {
    const int lpr = TC <= 8 ? 8 : TC <= 16 ? 16 : TC <= 32 ? 32 : 64;
    const int j = q / lpr, c = q - j * lpr;
    const int r = lpr == 8 ? (j >> 1) + 4 * (j & 1) : j;         // LPR = 8: rows r, r + 4 share a 16-lane group
    const bool ok = r < TR && c < TC;
    *prow = ok ? r : 0x4000;
    *pcol = c;
    *pp0 = (r < TR ? 2 * r * PW : 0) + c;                          // (idle lanes read inside the plane, next to their row's pixels)
  } else {
    const bool ok = q < TR * TC;
    const int r = q / TC, c = q - r * TC;
    *prow = ok ? r : 0x4000;
    *pcol = c;
    *pp0 = ok ? r * PW + c : 0;
  }
}
// Which 16-byte half of its 32-byte plane entry holds k 0-7 of patch position pp (row r = pp / PW, place q = pp % PW in the row):
// half (k >> 3) ^ swap.  A ds_read_b128 is served in four groups of 16 lanes -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
// the same + 32 (MI355X_MICROARCH.md, LDS) -- one LDS cycle per group when its 16 lanes hit 16 different 16-byte slots of the
// 256-byte bank row; slot = 2 (pp mod 8) + half ^ swap, so two positions of a group that agree mod 8 must differ in `swap`.
// Round 2-4 swapped on bit 3 of pp, which is conflict-free for 16 CONSECUTIVE positions per group -- not what the hardware's
// groups read: every fragment read of the config-5 tiles was 2-way conflicted (SQ_LDS_BANK_CONFLICT 65 % / 49 % of the LDS
// cycles, profiles/r04_bf16c5_pmc_sq1.txt; the model below says 50 %), at one read per MFMA exactly the matrix pipe's time.
// mode 0: bit 3 of pp; 1: bit 1 of q; 2: bit 0 of r; 3: bit 1 of r.  The host picks the mode with the fewest conflict cycles.
__host__ __device__ inline int ws128_swap(int mode, int pp, int PW) {
  const int r = pp / PW, q = pp - r * PW;
  return (mode == 1 ? (q >> 1) : mode == 2 ? r : mode == 3 ? (r >> 1) : (pp >> 3)) & 1;
}
// LDS cycles of the fragment reads of one tile (all M-tiles x 9 taps, one k16 plane) under swap mode `mode`; 4 per read = conflict-free.
static int ws128_read_cycles(bool s2, int TR, int TC, int mode) {
  static const int kGroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  const int PW = s2 ? 2 * TC + 1 : TC + 2, kMT = s2 ? 2 : 4;
  int cycles = 0;
  for (int mt = 0; mt < kMT; ++mt)
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      for (int half = 0; half < 2; ++half)
        for (int g = 0; g < 2; ++g) {
          int addr[16], worst = 1;
          for (int i = 0; i < 16; ++i) {
            int prow, pcol, pp0;
            ws128_lane_pixel(s2, TR, TC, PW, mt * 32 + kGroup[g][i], &prow, &pcol, &pp0);
            const int pp = pp0 + ky * PW + (s2 ? (kx == 1 ? TC + 1 : kx >> 1) : kx);
            addr[i] = pp * 2 + (half ^ ws128_swap(mode, pp, PW));      // in 16-byte units
          }
          for (int i = 0; i < 16; ++i) {     // distinct addresses on the same slot (equal addresses broadcast)
            int ways = 0;
            for (int k = 0; k < 16; ++k) {
              bool first = (addr[k] & 15) == (addr[i] & 15);
              for (int m = 0; first && m < k; ++m) first = addr[m] != addr[k];
              ways += first ? 1 : 0;
            }
            worst = ways > worst ? ways : worst;
          }
          cycles += worst;
        }
    }
  return cycles;
}
static int ws128_best_swap(bool s2, int TR, int TC) {
#ifdef TSM_WS128_SWZ_FORCE      // A/B builds only (TSM_BUILD_DEFS): 0 = rounds 2-4's bit-3 swap
  return TSM_WS128_SWZ_FORCE;
#endif
  int best = 0, best_c = ws128_read_cycles(s2, TR, TC, 0);
  for (int mode = 1; mode < 4; ++mode) {
    const int c = ws128_read_cycles(s2, TR, TC, mode);
    if (c < best_c) { best = mode; best_c = c; }
  }
  return best;
}

// The stride-1 tile of a frame: fewest tiles first (ws_tile_geometry's rule), and among the shapes with that many tiles the one whose
// fragment reads cost the fewest LDS cycles under its best swap -- the shapes differ by a factor of two there: on 32 x 32 frames
// 16 x 8 and 8 x 16 both give 8 tiles, but an M-tile of four 8-pixel rows puts FOUR lanes of a 16-lane read group on one pair of
// 16-byte slots (two entries per slot pair: 2-way conflicts whatever the swap, 50 % of the LDS cycles), while two 16-pixel rows
// with the halves swapped on the row's parity read conflict-free.
static bool ws128_tile_geometry(int H, int W, int *tr_out, int *tc_out, int *swz_out) {
  // (a pure function of the frame size, but ~10^7 operations of modelling: every launch of an engine asks for the same one or
  //  two sizes, so the calling thread remembers its last four answers)
  struct Memo { int H, W, tr, tc, swz; bool ok; };
  static thread_local Memo memo[4] = {};
  static thread_local int memo_next = 0;
  for (const Memo &m : memo)
    if (m.H == H && m.W == W && m.H > 0) {
      *tr_out = m.tr; *tc_out = m.tc;
      if (swz_out) *swz_out = m.swz;
      return m.ok;
    }
  long best_tiles = -1;
  int best_tr = 0, best_tc = 0, best_cycles = 0, best_swz = 0;
  for (int tc = 4; tc <= 128; ++tc) {
    int tr = 128 / tc;
    if (tr > H) tr = H;
    if (tr < 1 || (tr + 2) * (tc + 2) > kW8PatchMax) continue;
    const long tiles = (long)((H + tr - 1) / tr) * ((W + tc - 1) / tc);
    if (best_tiles >= 0 && tiles > best_tiles) continue;
    const int swz = ws128_best_swap(false, tr, tc), cycles = ws128_read_cycles(false, tr, tc, swz);
    if (best_tiles < 0 || tiles < best_tiles || cycles < best_cycles) {
      best_tiles = tiles; best_tr = tr; best_tc = tc; best_cycles = cycles; best_swz = swz;
    }
  }
  *tr_out = best_tr;
  *tc_out = best_tc;
  if (swz_out) *swz_out = best_swz;
  memo[memo_next] = Memo{H, W, best_tr, best_tc, best_swz, best_tiles > 0};
  memo_next = (memo_next + 1) & 3;
  return best_tiles > 0;
}

template <bool S2>
__global__ void __launch_bounds__(256, 1) conv3x3_ws128_kernel(const WsParams p) {
  constexpr int kRounds = S2 ? kS2Rounds : kW8Rounds, kPlane = S2 ? kS2Plane : kW8Plane, kBuf = 8 * kPlane;
  constexpr int kMT = S2 ? 2 : 4;                  // M-tiles of a tile
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kBuf | bias
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W, TR = p.tr, TC = p.tc;            // (H, W: the INPUT frame)
  const int Ho = S2 ? (H - 1) / 2 + 1 : H, Wo = S2 ? (W - 1) / 2 + 1 : W;
  const int PW = S2 ? 2 * TC + 1 : TC + 2, PH = S2 ? 2 * TR + 1 : TR + 2;
  const int nr = (PH * PW + 31) >> 5;
  const int tiles_x = (Wo + TC - 1) / TC, tiles_y = (Ho + TR - 1) / TR, tiles_f = tiles_x * tiles_y;
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
  float *bias_lds = reinterpret_cast<float *>(lds + 2 * kBuf);
  if (tid < 128) bias_lds[tid] = p.bias2[tid];
  const float floor_ = p.relu ? 0.f : -INFINITY;

  // ---- loader: plane g holds bytes [32 g, 32 g + 32) of every patch position (halves swapped where (position >> 3) is
  // odd); in round i this lane fills half (lane & 1) of position 32 i + (lane >> 1), in planes 2 wave and 2 wave + 1.
  // Position -> patch pixel: row-major; S2: within a row the even columns first, then the odd ones.
  // (which of the pixel's two 16-byte chunks of a plane this lane fetches: the half it fills, swapped where ws128_swap says so)
  unsigned dslot[kRounds];                         // (byte offset of the lane's chunk relative to the patch origin's chunk 4 wave) >> 4 | patch column << 24
#pragma unroll
  for (int i = 0; i < kRounds; ++i) {
    const int pidx = 32 * i + (lane >> 1);
    const int pr = pidx / PW, q = pidx - pr * PW;
    const int pc = S2 ? (q <= TC ? 2 * q : 2 * (q - TC - 1) + 1) : q;
    const int hsel = (lane & 1) ^ ws128_swap(p.swz, pidx, PW);
    dslot[i] = (unsigned)((pr * W + pc) * 16 + hsel) | ((unsigned)pc << 24);
  }
  // The patch of a tile, as 2 x kRounds LDS-DMA operations that can be issued one at a time: in a burst behind the barrier a
  // piece costs ~100 cycles of issue (in-kernel stamps of front_s2_kernel, round 5) -- 12 / 20 of them per tile against 144 / 72
  // MFMAs of 32 cycles --, behind an MFMA about nothing.  Operation k = round k >> 1 of plane 2 wave + (k & 1).
  struct PatchDma {
    const char *base;      // the tile's frame
    int tbase, x0;
    unsigned char *dst;
    bool live;
  };
  auto prep_patch = [&](int t, int b, bool live) -> PatchDma {
    PatchDma d;
    const int f = t / tiles_f, rem = t - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = S2 ? 2 * ty * TR - 1 : ty * TR - 1;                  // the patch origin
    d.x0 = S2 ? 2 * tx * TC - 1 : tx * TC - 1;
    d.base = reinterpret_cast<const char *>(p.x) + (size_t)f * frame_bytes;
    d.tbase = (y0 * W + d.x0) * 256 + 4 * wave * 16;
    d.dst = lds + b * kBuf + 2 * wave * kPlane;
    d.live = live;
    return d;
  };
  auto patch_piece = [&](const PatchDma &d, int k) {
    const int i = k >> 1, pl = k & 1;
    if (d.live && i < nr && (!S2 || i < kS2Rounds - 1 || lane < 2)) {   // (S2's tenth round: position 288 alone)
      const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(d.base), 0, frame_bytes, 0x00020000);
      const int xg = d.x0 + (int)(dslot[i] >> 24);
      const unsigned off = (unsigned)d.tbase + ((dslot[i] & 0xFFFFFFu) << 4) + (pl ? 32u : 0u);
      const unsigned o = (unsigned)xg < (unsigned)W ? off : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(d.dst + pl * kPlane + i * 1024), 16, (int)o, 0, 0, 0);
    }
  };
  auto issue_patch = [&](int t, int b) {
    const PatchDma d = prep_patch(t, b, true);
#pragma unroll
    for (int k = 0; k < 2 * kRounds; ++k) patch_piece(d, k);
  };

  // ---- this lane's pixel in each of the M-tiles (shared by all waves)
  int prow[kMT], pcol[kMT], pp0[kMT];
  unsigned hsw = 0;                                // bit 9 mt + tap: the 16-byte half of its plane entry this lane reads at that tap
#pragma unroll
  for (int mt = 0; mt < kMT; ++mt) {
    ws128_lane_pixel(S2, TR, TC, PW, mt * 32 + l31, &prow[mt], &pcol[mt], &pp0[mt]);
    if (mt < 3) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int pp = pp0[mt] + ky * PW + (S2 ? (kx == 1 ? TC + 1 : kx >> 1) : kx);
        hsw |= (unsigned)(half ^ ws128_swap(p.swz, pp, PW)) << (9 * mt + tap);
      }
    }
  }
  unsigned hsw3 = 0;                               // ... M-tile 3 (stride 1 only)
  if constexpr (!S2) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      hsw3 |= (unsigned)(half ^ ws128_swap(p.swz, pp0[3] + ky * PW + kx, PW)) << tap;
    }
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
            o, rsrcYall, (int)(yoff[m] == kInvalid ? kInvalid : yoff[m] + (unsigned)(wave * 64 + (2 * half + qq) * 16)), 0, TSM_AUX_WS);
      }
    }
  };
  auto out_off = [&](int tt, int mt) -> unsigned {
    const int f = tt / tiles_f, rem = tt - f * tiles_f;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy = ty * TR + prow[mt], ox = tx * TC + pcol[mt];
    return (oy < Ho && ox < Wo) ? (unsigned)(((f * Ho + oy) * Wo + ox) * 256) : kInvalid;
  };

  // A pair of M-tiles (2 mp, 2 mp + 1): 72 steps (tap, g) of two pixel-fragment reads (two steps ahead) and two MFMAs
  // with the same weight fragment; the pieces of the previous pair are spread over steps 3, 10, .., 66.
  // (`dma`: the next tile's patch rides on this pair's first steps -- all of its operations in FRONT of the step that issues the
  //  previous pair's first stores, so that the counted wait at the tile's end still leaves exactly that many stores in flight)
  auto mpair = [&](const unsigned char *buf, int mp, f32x16 (&acc)[2], const f32x16 (&prev)[2], const unsigned (&prev_off)[2], bool carry,
                   const PatchDma &dma) {
    u32x4 px[4][2];
    unsigned tb[2] = {0u, 0u};
    auto rd = [&](int s) {
      const int tap = s >> 3, g = s & 7, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (g == 0) {
          const int mt = 2 * mp + m;
          const int pp = pp0[mt] + ky * PW + (S2 ? (kx == 1 ? TC + 1 : kx >> 1) : kx);
          const unsigned hb = mt < 3 ? (hsw >> (9 * mt + tap)) & 1u : (hsw3 >> tap) & 1u;
          tb[m] = (unsigned)(pp * 32) + (hb << 4);
        }
        px[s & 3][m] = *reinterpret_cast<const u32x4 *>(buf + tb[m] + g * kPlane);
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
      for (int m = 0; m < 2; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wr[s]), __builtin_bit_cast(bf16x8, px[s & 3][m]),
                                                         acc[m], 0, 0, 0);
        // stride 1: an operation behind the first MFMA of steps 4, 6, .., 26; stride 2: behind both MFMAs of steps 4 .. 13 -- all
        // before step 31, where the previous pair's first two stores are issued
        if constexpr (S2) {
          if constexpr (s >= 4 && s < 4 + kRounds) {
            if (carry) patch_piece(dma, 2 * (s - 4) + m);
          }
        } else {
          if constexpr (s >= 4 && s < 4 + 4 * kRounds && (s & 1) == 0) {
            if (carry && m == 0) patch_piece(dma, (s - 4) >> 1);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  f32x16 accA[2], accB[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[m][e] = 0.f;
  unsigned offA[2], offB[2] = {kInvalid, kInvalid};                     // nothing to store before the first tile
  // virtual tile index -> tile: XCD-chunked (neighbouring tiles share halo rows and columns of their patches), then the
  // engine's alternating direction
  auto tile_of = [&](int v) {
    const int c = (int)xcd_chunked(v, ntiles);
    return p.reverse ? ntiles - 1 - c : c;
  };
  int t = blockIdx.x, nb = 0;
  if (t < ntiles) issue_patch(tile_of(t), 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (S2) {
    // one pair per tile: the accumulator sets alternate from tile to tile (the epilogue of tile i rides on tile i + 1)
    auto step = [&](f32x16 (&cur)[2], const f32x16 (&prev)[2], const unsigned (&prev_off)[2], unsigned (&cur_off)[2]) {
      __builtin_amdgcn_s_barrier();    // every wave's share of this patch has landed; nobody still reads the other buffer
      const int tn = t + gridDim.x;
      const PatchDma dma = prep_patch(tile_of(tn < ntiles ? tn : t), nb ^ 1, tn < ntiles);
      const int tt = tile_of(t);
      mpair(lds + nb * kBuf, 0, cur, prev, prev_off, true, dma);
      cur_off[0] = out_off(tt, 0); cur_off[1] = out_off(tt, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                  // the next patch is older than this tile's four stores
      t = tn;
      nb ^= 1;
    };
    for (;;) {
      step(accA, accB, offB, offA);
      if (t >= ntiles) {
#pragma unroll
        for (int k = 0; k < 10; ++k) epi_piece(accA, k, offA);
        break;
      }
      step(accB, accA, offA, offB);
      if (t >= ntiles) {
#pragma unroll
        for (int k = 0; k < 10; ++k) epi_piece(accB, k, offB);
        break;
      }
    }
  } else {
    for (; t < ntiles; t += gridDim.x, nb ^= 1) {
      __builtin_amdgcn_s_barrier();      // every wave's share of this patch has landed; nobody still reads the other buffer
      const int tn = t + gridDim.x;
      const PatchDma dma = prep_patch(tile_of(tn < ntiles ? tn : t), nb ^ 1, tn < ntiles);
      const int tt = tile_of(t);
      const unsigned char *buf = lds + nb * kBuf;
      mpair(buf, 0, accA, accB, offB, true, dma);                         // (B = M-tiles 2, 3 of the previous tile)
      offA[0] = out_off(tt, 0); offA[1] = out_off(tt, 1);
      mpair(buf, 1, accB, accA, offA, false, dma);
      offB[0] = out_off(tt, 2); offB[1] = out_off(tt, 3);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                    // the next patch is older than this iteration's eight stores
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) epi_piece(accB, k, offB);
  }
}

static bool conv3x3_ws128_common(const ConvParams &p) {
  return p.prec == kPrecBf16 && p.C == 128 && p.Cout == 128 && p.Kp == 1152 && p.pad == 1 && !p.res && !p.x2 && p.T == 0 &&
         p.kseg_len == 0 && (double)p.M * 256.0 < 2.0e9 && (double)p.Hi * p.Wi * 256.0 < 2.0e9;
}

// stride 2 (layer2.0's conv2)
static bool conv3x3_ws128s2_valid(const ConvParams &p) {
  int tr, tc;
  return conv3x3_ws128_common(p) && p.stride == 2 && p.Ho == (p.Hi - 1) / 2 + 1 && p.Wo == (p.Wi - 1) / 2 + 1 && p.Wi <= 2048 &&
         ws_s2_tile_geometry(p.Ho, p.Wo, &tr, &tc);
}

bool conv3x3_ws128_valid(const ConvParams &p) {
  int tr, tc;
  if (conv3x3_ws128s2_valid(p)) return true;
  return conv3x3_ws128_common(p) && p.stride == 1 && p.Hi == p.Ho && p.Wi == p.Wo && ws128_tile_geometry(p.Hi, p.Wi, &tr, &tc, nullptr);
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
                                               0, TSM_AUX_WS);
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

// NSPLIT = 2: the launch computes 2 * COUT output channels, one half per workgroup.  Workgroups b and b + 8 -- neighbours on ONE XCD, so
// the second read of a pixel tile comes from that XCD's L2 -- walk the same pixel tiles in the same order, each with its half of
// the weights in registers (conv3 + downsample of layer2.0: 128 + 256 -> 512 channels, 2 x 48 fragments per wave).
template <int CIN, int COUT, bool DUAL, int NSPLIT = 1>
__global__ void __launch_bounds__(256, 1) conv1x1_wsn_kernel(const WsnParams p) {
  static_assert(NSPLIT == 1 || NSPLIT == 2, "one or two output-channel halves");
  constexpr int CTOT = COUT * NSPLIT;    // channels of an output row
  constexpr int PX = CIN <= 256 ? 128 : 64;
  constexpr int MT = PX / 32;
  constexpr int NG = CIN / 16;
  constexpr int NTW = COUT / 128;        // output-channel tiles per wave
  constexpr int PPW = NG / 4;            // planes filled per wave
  constexpr int kPlane = PX * 32;
  constexpr int kBuf = NG * kPlane;      // 64 KB
  constexpr int kAgpr = NTW * NG > 40 ? NTW * NG - 24 : 0;   // fragments pinned to accumulation registers (the forms with 48 / 64 of them)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x kBuf | bias
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntiles = (p.M + PX - 1) / PX;

  const int nh = NSPLIT == 2 ? ((int)blockIdx.x >> 3) & 1 : 0;          // this workgroup's half of the output channels
  const int tfirst = NSPLIT == 2 ? ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 4) : (int)blockIdx.x;
  const int tstep = NSPLIT == 2 ? (int)gridDim.x >> 1 : (int)gridDim.x;   // (NSPLIT = 2: the grid is a multiple of 16)
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, CTOT * CIN * 2, 0x00020000);
  u32x4 wr[NTW][NG];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      wr[nt][g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, ((nh * COUT + (wave * NTW + nt) * 32 + l31) * CIN + g * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (nt * NG + g < kAgpr) asm volatile("" : "+a"(wr[nt][g]));
  float *bias_lds = reinterpret_cast<float *>(lds + 2 * kBuf);
  if (tid < COUT) bias_lds[tid] = p.bias[nh * COUT + tid];
  const float floor_ = p.relu ? 0.f : -INFINITY;

  const int hsel = (lane & 1) ^ ((lane >> 4) & 1);
  const int C1 = DUAL ? p.K1 : CIN, C2 = CIN - C1;
  const int frame_bytes = p.HW * C1 * 2;
  const int frame2_bytes = DUAL ? p.Hi2 * p.Wi2 * C2 * 2 : 0;
  // The next tile's operands: prep() computes the tile's windows (scalar) and each lane's row offsets for its MT rounds;
  // piece(d, b, i, k) issues ONE 1-KiB LDS-DMA instruction (round i, plane PPW wave + k).  The compute loop issues a tile's
  // MT * PPW pieces one every fourth MFMA step -- a piece in a burst behind the barrier costs ~100 cycles of issue, one behind
  // an MFMA next to nothing (one wave per SIMD: nothing else would hide it) -- and tile 0's all at once.
  struct TileDma {
    const char *px, *px2;
    int szx, szx2;
    unsigned own[MT], own2[MT], vmask[MT];     // vmask: bit 0 row valid, bit 1 frame t + 1 exists, bit 2 frame t - 1 exists
  };
  auto prep = [&](int t, bool live) {
    TileDma d;
    const int m0 = t * PX;
    const long base_row = (long)m0 - p.HW;
    const size_t span = (size_t)(PX + 2 * p.HW) * C1 * 2;
    d.px = reinterpret_cast<const char *>(p.x) + base_row * (long)(C1 * 2);
    d.szx = (int)(span > 0x7FFFFFF0u ? 0x7FFFFFF0u : span);
    const int n0 = m0 / p.HW;                                           // first frame of the tile
    d.px2 = reinterpret_cast<const char *>(DUAL ? p.x2 : p.x) + (size_t)n0 * frame2_bytes;
    d.szx2 = DUAL ? (int)((size_t)(PX / p.HW + 2) * frame2_bytes > 0x7FFFFFF0u ? 0x7FFFFFF0u : (size_t)(PX / p.HW + 2) * frame2_bytes) : 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int pl = 32 * i + (lane >> 1);
      const int m = m0 + pl;
      const bool ok = live && m < p.M;
      const int n = (ok ? m : m0) / p.HW;
      const int tt = p.T > 0 ? n % p.T : 0;
      d.own[i] = (unsigned)((pl + p.HW) * C1 * 2);
      d.own2[i] = 0;
      if (DUAL) {
        const int rem = (ok ? m : m0) - n * p.HW;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        d.own2[i] = (unsigned)((n - n0) * frame2_bytes + ((oy * p.stride2) * p.Wi2 + ox * p.stride2) * C2 * 2);
      }
      d.vmask[i] = (ok ? 1u : 0u) | ((ok && tt < p.T - 1) ? 2u : 0u) | ((ok && tt > 0) ? 4u : 0u);
    }
    return d;
  };
  auto piece = [&](const TileDma &d, int b, int i, int k) {
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(d.px), 0, d.szx, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(d.px2), 0, d.szx2, 0x00020000);
    const int g = PPW * wave + k;
    const int c0 = (2 * g + hsel) * 8;
    lds_void *dst = (lds_void *)(lds + b * kBuf + g * kPlane + i * 1024);
    const bool ok = (d.vmask[i] & 1u) != 0;
    if (DUAL && c0 >= C1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX2, dst, 16, (int)(ok ? d.own2[i] + (unsigned)((c0 - C1) * 2) : kInvalid), 0, 0, 0);
    } else {
      unsigned off = d.own[i];
      bool valid = ok;
      if (p.T > 0 && c0 < p.fold) { off = d.own[i] + (unsigned)frame_bytes; valid = (d.vmask[i] & 2u) != 0; }
      else if (p.T > 0 && c0 < 2 * p.fold) { off = d.own[i] - (unsigned)frame_bytes; valid = (d.vmask[i] & 4u) != 0; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, dst, 16, (int)(valid ? off + (unsigned)(c0 * 2) : kInvalid), 0, 0, 0);
    }
  };
  constexpr int kPieces = MT * PPW, kEvery = NG / PPW;   // one piece per kEvery (= 4) MFMA steps: the last one behind step MT * NG - kEvery
  static_assert(kEvery * PPW == NG && kPieces * kEvery == MT * NG, "the pieces spread evenly over the tile's MFMA steps");

  const size_t ybytes = (size_t)p.M * CTOT * 2;
  int t = tfirst, nb = 0;
  if (t < ntiles) {
    const TileDma d0 = prep(p.reverse ? ntiles - 1 - t : t, true);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int k = 0; k < PPW; ++k) piece(d0, 0, i, k);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  for (; t < ntiles; t += tstep, nb ^= 1) {
    __builtin_amdgcn_s_barrier();
    const int tn = t + tstep;
    const int tnn = tn < ntiles ? tn : t;                                // (no next tile: dead pieces, zeros, no memory traffic)
    const TileDma dn = prep(p.reverse ? ntiles - 1 - tnn : tnn, tn < ntiles);
    const int tt = p.reverse ? ntiles - 1 - t : t;
    const unsigned char *buf = lds + nb * kBuf;
    // output window of the tile (rebased: 32-bit offsets whatever M * COUT is)
    const size_t y0 = (size_t)tt * PX * CTOT * 2;
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.y) + y0, 0, (int)(ybytes - y0 > (size_t)PX * CTOT * 2 ? (size_t)PX * CTOT * 2 : ybytes - y0), 0x00020000);
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
        if ((mt * NG + g) % kEvery == kEvery - 1) {                      // the next tile's piece (mt * NG + g) / kEvery, behind this step's MFMAs
          const int j = (mt * NG + g) / kEvery;
          piece(dn, nb ^ 1, j / PPW, j % PPW);
        }
      }
      const unsigned yoff = (unsigned)(pp * CTOT * 2 + nh * COUT * 2);   // (rows past M fall outside the rebased window)
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
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)(yoff + (unsigned)(ch0 * 2 + (2 * half + qq) * 16)), 0, TSM_AUX_WS);
        }
      }
    }
    wait_vmcnt(NTW * 2);        // the next tile's last piece is older than the last M-tile's stores (and younger than every other store)
  }
}

bool conv1x1_wsn_valid(const ConvParams &p) {
  if (p.prec != kPrecBf16 || p.stride != 1 || p.pad != 0 || p.Hi != p.Ho || p.Wi != p.Wo || p.res || p.kseg_len != 0) return false;
  if ((double)(128 + 2.0 * p.Hi * p.Wi) * p.C * 2.0 >= 2.0e9) return false;
  if (p.x2) {   // conv3 + downsample of layer1.0: 64 + 64 -> 256; of layer2.0: 128 + 256 -> 512 (two halves of 256)
    const bool l1 = p.C == 64 && p.C2 == 64 && p.K1 == 64 && p.Kp == 128 && p.Cout == 256;
    const bool l2 = p.C == 128 && p.C2 == 256 && p.K1 == 128 && p.Kp == 384 && p.Cout == 512 && device_info().n_cu >= 16;
    return p.T == 0 && (l1 || l2) && (double)(128.0 / (p.Hi * p.Wi) + 2.0) * p.Hi2 * p.Wi2 * p.C2 * 2.0 < 2.0e9;
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


static int ws_grid_setup() { return device_info().n_cu; }

hipError_t launch_conv3x3_ws(ConvParams p, hipStream_t s) {
  if (conv3x3_ws128s2_valid(p)) {
    WsParams q{};
    q.x = p.x; q.w2 = p.w; q.bias2 = p.bias; q.y = p.y;
    q.N = p.N; q.H = p.Hi; q.W = p.Wi; q.M = p.M; q.relu = p.relu; q.reverse = p.reverse;
    ws_s2_tile_geometry(p.Ho, p.Wo, &q.tr, &q.tc);
    {
      static thread_local int m_tr = 0, m_tc = 0, m_swz = 0;       // (the same modelling cost: remembered per thread)
      if (m_tr != q.tr || m_tc != q.tc) { m_swz = ws128_best_swap(true, q.tr, q.tc); m_tr = q.tr; m_tc = q.tc; }
      q.swz = m_swz;
    }
    const long ntiles = (long)q.N * ((p.Ho + q.tr - 1) / q.tr) * ((p.Wo + q.tc - 1) / q.tc);
    const int n_cu = ws_grid_setup();
    if (device_info().status != hipSuccess) return device_info().status;
    TSM_KLAUNCH(conv3x3_ws128_kernel<true>, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kS2LdsBytes, s, q);
    return hipGetLastError();
  }
  if (conv3x3_ws128_valid(p)) {
    WsParams q{};
    q.x = p.x; q.w2 = p.w; q.bias2 = p.bias; q.y = p.y;
    q.N = p.N; q.H = p.Hi; q.W = p.Wi; q.M = p.M; q.relu = p.relu; q.reverse = p.reverse;
    ws128_tile_geometry(q.H, q.W, &q.tr, &q.tc, &q.swz);
    const long ntiles = (long)q.N * ((q.H + q.tr - 1) / q.tr) * ((q.W + q.tc - 1) / q.tc);
    const int n_cu = ws_grid_setup();
    if (device_info().status != hipSuccess) return device_info().status;
    TSM_KLAUNCH(conv3x3_ws128_kernel<false>, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kW8LdsBytes, s, q);
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
  TSM_KLAUNCH(conv3x3_ws_kernel<false>, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kWsLdsBytes, s, q);
  return hipGetLastError();
}

hipError_t launch_conv1x1_ws(const ConvParams &p, hipStream_t s) {
  if (!conv1x1_ws_valid(p)) return hipErrorInvalidValue;
  Ws1Params q{};
  q.x = p.x; q.w = p.w; q.bias = p.bias; q.y = p.y;
  q.M = p.M; q.HW = p.Hi * p.Wi; q.T = p.T; q.fold = p.fold; q.relu = p.relu; q.reverse = p.reverse;
  const int n_cu = ws_grid_setup();
  if (device_info().status != hipSuccess) return device_info().status;
  const int ntiles = (p.M + 127) / 128;
  const unsigned grid = (unsigned)(ntiles < n_cu ? ntiles : n_cu);
  if (p.C == 256) TSM_KLAUNCH(conv1x1_ws_kernel<256>, dim3(grid), dim3(256), 2 * 16 * 4096 + 256, s, q);
  else TSM_KLAUNCH(conv1x1_ws_kernel<64>, dim3(grid), dim3(256), 2 * 4 * 4096 + 256, s, q);
  return hipGetLastError();
}

hipError_t launch_conv1x1_wsn(const ConvParams &p, hipStream_t s) {
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
  if (p.x2 && p.Kp == 384) {
    const int pairs = n_cu >> 4 << 3;                                     // workgroup pairs: a multiple of 8 (b and b + 8 share an XCD)
    const int np = ntiles < pairs ? (ntiles + 7) / 8 * 8 : pairs;         // (a pair without a tile leaves at once)
    TSM_KLAUNCH((conv1x1_wsn_kernel<384, 256, true, 2>), dim3((unsigned)(2 * np)), block, 2 * 24 * 2048 + 1024, s, q);
  } else if (p.x2) TSM_KLAUNCH((conv1x1_wsn_kernel<128, 256, true>), grid, block, kLds, s, q);
  else if (p.C == 256) TSM_KLAUNCH((conv1x1_wsn_kernel<256, 128, false>), grid, block, kLds, s, q);
  else if (p.Cout == 128) TSM_KLAUNCH((conv1x1_wsn_kernel<512, 128, false>), grid, block, kLds, s, q);
  else TSM_KLAUNCH((conv1x1_wsn_kernel<512, 256, false>), grid, block, kLds, s, q);
  return hipGetLastError();
}

// Fused23Params with bf16 operands: w3f = conv3's packed weights [256][64] bf16 (row-major, as launch_conv takes them).
bool conv23_ws_valid(int n, int h, int w) {
  int tr, tc;
  return n > 0 && h > 0 && w > 0 && (double)h * w * 512.0 < 2.0e9 && (double)n * h * w < 2.0e9 && ws_tile_geometry(h, w, &tr, &tc);
}

hipError_t launch_conv23_ws(const Fused23Params &p, hipStream_t s) {
  if (!conv23_ws_valid(p.N, p.H, p.W) || p.kseg_len != 0) return hipErrorInvalidValue;
  WsParams q{};
  q.x = p.x; q.w2 = p.w2; q.bias2 = p.bias2; q.w3 = p.w3f; q.bias3 = p.bias3; q.res = p.res; q.y = p.y;
  q.N = p.N; q.H = p.H; q.W = p.W; q.M = p.M; q.relu = 1; q.reverse = p.reverse;
  ws_tile_geometry(q.H, q.W, &q.tr, &q.tc);
  const long ntiles = (long)q.N * ((q.H + q.tr - 1) / q.tr) * ((q.W + q.tc - 1) / q.tc);
  const int n_cu = ws_grid_setup();
  if (device_info().status != hipSuccess) return device_info().status;
  TSM_KLAUNCH(conv3x3_ws_kernel<true>, dim3((unsigned)(ntiles < n_cu ? ntiles : n_cu)), dim3(256), kWsLdsBytes3All, s, q);
  return hipGetLastError();
}

hipError_t opt_in_ws() {
  hipError_t first = hipSuccess;
  auto opt_in = [&](const void *fn, size_t bytes) {
    const hipError_t st = lds_opt_in(fn, bytes);
    if (st != hipSuccess && first == hipSuccess) first = st;
  };
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws_kernel<false>), kWsLdsBytes);
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws_kernel<true>), kWsLdsBytes3All);
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws128_kernel<false>), kW8LdsBytes);
  opt_in(reinterpret_cast<const void *>(&conv3x3_ws128_kernel<true>), kS2LdsBytes);
  opt_in(reinterpret_cast<const void *>(&conv1x1_ws_kernel<256>), 2 * 16 * 4096 + 256);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<128, 256, true>), 2 * 65536 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<384, 256, true, 2>), 2 * 24 * 2048 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<256, 128, false>), 2 * 65536 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<512, 128, false>), 2 * 65536 + 1024);
  opt_in(reinterpret_cast<const void *>(&conv1x1_wsn_kernel<512, 256, false>), 2 * 65536 + 1024);
  return first;
}

}  // namespace tsm
