// conv31_fused_kernel: Bottleneck.conv3 + bn3 + residual + ReLU of block b  AND  temporal shift + conv1 + bn1 + ReLU of
// block b + 1 in ONE launch (bf16; layer2 / layer3 blocks without a downsample branch).
#include "tsm_device.h"

namespace tsm {

// ---------------------------------------------------------------------------------------------
// Why: in the bf16 engine every launch of layer2 sits on its own roofline, so what is left are the bytes BETWEEN
// launches: conv3 of block b writes the block output y (H*W*C*2 bytes per frame) and conv1 of block b + 1 reads all of it
// back one launch later (1.07 GB per block at the config-5 size).  Fusing the two across the block boundary looked
// forbidden -- conv1 reads its first C/8 channels from frame t + 1 and the next C/8 from frame t - 1 (models/tsm.py:35-50),
// so a SPATIAL tile of y does not hold conv1's input.  A CLIP-MAJOR tile does: a tile here is ALL T frames of one clip x
// PX consecutive pixels (T * PX = 256 or 128 rows, row = t * PX + px), and the shifted channels of row (t, px) are the same
// channels of rows (t +- 1, px) of the SAME tile -- a row offset of +- PX inside the tile, zeros where t +- 1 leaves the clip.
//
// One persistent 8-wave workgroup per CU (two waves per SIMD, <= 256 registers each).  CH = 1 (N1 = 128): tiles of 256
// rows, wave w owns rows 32 w .. 32 w + 31 in both GEMMs.  CH = 2 (N1 = 256: GEMM2's 32 x 256 accumulator does not fit one
// wave): tiles of 128 rows, the two waves of a PAIR share 32 rows and split the columns of both GEMMs (GEMM1: 32 of the
// chunk's 64 channels each, GEMM2: 128 of the 256 output channels each); the chunk's epilogue is split by rows, 16 each.
// Per tile:
//   * t2 (conv3's input, [rows][K3] bf16) is read ONCE into registers as the A fragments of GEMM1 (K3 / 16 fragments per
//     lane): CH = 1 through an LDS staging buffer filled by LDS-DMA a whole tile ahead; CH = 2 (no LDS left for that) by
//     16-byte loads issued right after the last GEMM1 of the previous tile, under that chunk's epilogue and GEMM2;
//   * the block channels are walked in chunks of 64:  GEMM1  y[:, chunk] = t2 * W3[chunk, :]^T  (16 MFMAs per wave, B
//     fragments from the chunk of W3 that LDS-DMA brought in during the previous chunk);  epilogue through an fp32 LDS slab
//     (CH = 1: wave-private [8][68] sub-slabs; CH = 2: the pair's [32][68]): + bias3, + residual (16 bytes per lane = whole 128-byte row segments per 8 lanes, loaded into
//     registers one chunk ahead), ReLU, bf16 -> (a) stored to y (the next block's identity needs it: written once, never
//     re-read by this kernel), (b) written to a [256][64] bf16 LDS tile;  GEMM2  t1 += shift(y[:, chunk]) * W1[:, chunk]^T
//     (16 MFMAs per wave; the A fragments are rows r, r + PX or r - PX of the LDS tile according to the chunk's place in
//     the channel order, a zero row past the clip's ends; the B fragments the chunk of W1 brought in by LDS-DMA);
//   * after the last chunk: t1 = relu(acc + bias1) -> bf16 -> stored ([256][N1], conv2 of block b + 1 reads it).
// HBM bytes per tile row: K3*2 (t2) + C*2 (residual) + C*2 (y) + N1*2 (t1) against + C*2 more for the two launches it
// replaces; the weights (C*K3 + N1*C elements per tile) stream from L2.  Three barriers per chunk; every vector-memory
// wait is a counted vmcnt (a running count of issued operations against the count at the awaited operation's issue) over
// the issue order of a chunk
//     [W1 chunk: NW1 DMA | W3 next chunk: NW3 DMA | CH = 2, last chunk: the next tile's A fragments |
//      (store y, load next residual) x 4 / CH | CH = 1: t2 of the next tile, P DMA],
// never vmcnt(0) inside the loop: the residual stream, the weight stream and the next tile's t2 stay in flight under
// both GEMMs and the epilogue.
// Products enter every accumulator in the separate kernels' order (k16 groups ascending from a zero accumulator) and the
// two epilogues are theirs (conv_bf16_256p's residual arm; its shifted-conv1 arm): bit-identical to the two launches.
// Needs T | rows with 8 <= rows / T (all frames of a clip in one tile), fold % 64 == 0 (a chunk is shifted as a whole).
// ---------------------------------------------------------------------------------------------
template <int K3, int C, int N1, int CH> struct C31 {
  static constexpr int NW = 8;                     // waves
  static constexpr int M = 32 * NW / CH;           // tile rows = T frames x PX pixels
  static constexpr int NT = 64 * NW;               // threads
  static constexpr bool STAGE = CH == 1;           // t2 through an LDS staging buffer (else straight into registers)
  static constexpr int KT1 = K3 / 16;              // k16 steps of GEMM1
  static constexpr int NC = C / 64;                // chunks of the block's channels
  static constexpr int NTL1 = 2 / CH;              // GEMM1 N-tiles per wave (of the chunk's two)
  static constexpr int NTL2 = N1 / 32 / CH;        // GEMM2 N-tiles per wave
  static constexpr int NQ = 4 / CH;                // 8-row epilogue steps per wave and chunk
  static constexpr int RD = CH;                    // chunks the residual is requested ahead: a tile of 128 rows turns a chunk over in
                                                   // less than an HBM round trip, so CH = 2 keeps TWO chunks of residual in flight
  static constexpr int RB3 = K3 * 2;               // bytes per row of t2 / of W3
  static constexpr int LPR3 = RB3 / 16;            // lanes (16-byte slots) per such row
  static constexpr int RPP3 = 1024 / RB3;          // rows per 1-KiB DMA piece
  static constexpr int NW3 = 64 * RB3 / 1024 / NW; // DMA pieces per wave: a chunk of W3 (64 rows)
  static constexpr int NW1 = N1 * 128 / 1024 / NW; // ... a chunk of W1 (N1 rows x 64 channels)
  static constexpr int NT2 = STAGE ? M * RB3 / 1024 / NW : 0;   // ... the staged t2 tile (the wave's own 32 rows)
  static constexpr int PT2 = STAGE ? 2 * NT2 / NC : 0;          // t2 pieces of the NEXT tile issued per chunk, in the first NC / 2 chunks
  static constexpr int AF = STAGE ? 0 : KT1;       // register loads of the next tile's A fragments, in a tile's last chunk
  static constexpr int NT1S = 4 * (N1 / 64) / CH;  // t1 stores per wave and tile
  static constexpr int LOG_C_N1 = C / N1 == 4 ? 2 : C / N1 == 2 ? 1 : 0;
  static constexpr int kSlabBytes = STAGE ? NW * 2176 : (NW / CH) * 8704;   // [8][68] per wave / [32][68] per pair, fp32
  static constexpr int kW3 = 0;
  static constexpr int kW1 = kW3 + 64 * RB3;
  static constexpr int kY = kW1 + N1 * 128;
  static constexpr int kT2 = kY + M * 128;
  static constexpr int kSlab = kT2 + (STAGE ? M * RB3 : 0);
  static constexpr int kBias3 = kSlab + kSlabBytes;
  static constexpr int kBias1 = kBias3 + C * 4;
  static constexpr int kZero = kBias1 + N1 * 4;
  static constexpr int kBytes = kZero + 128;
  // ---- counted waits: vector-memory operations of a wave, in issue order, per chunk c of a tile:
  //   W1(c) [NW1] | W3(c + 1) [NW3] | c last: the next tile's A fragments [AF] | (store y, load the residual of chunk c + RD) x NQ |
  //   t2 pieces of the next tile [P(c)] | c last: the tile's t1 stores [NT1S]
  // A wait names how many operations YOUNGER than the awaited one may stay in flight (they retire in order).
  static constexpr int P(int c) { return (STAGE && c < NC / 2) ? PT2 : 0; }
  static constexpr int tot(int c) { return NW1 + NW3 + 2 * NQ + P(c) + (c == NC - 1 ? AF + NT1S : 0); }
  // W3's chunk nc (issued in chunk nc - 1, behind that chunk's W1): the rest of that chunk
  static constexpr int wait_w3(int nc) { return nc == 0 ? AF + 2 * NQ + NT1S : 2 * NQ + P(nc - 1); }
  // W1's chunk nc (issued at the head of chunk nc): the rest of this chunk up to GEMM2
  static constexpr int wait_w1(int nc) { return NW3 + (nc == NC - 1 ? AF : 0) + 2 * NQ + P(nc); }
  // the residual of a step of chunk nc (issued at the same step of chunk nc - RD, behind its store): the rest of that chunk,
  // RD - 1 whole chunks, and this chunk up to the step -- the same number for every step
  static constexpr int wait_res(int nc) {
    const int c0 = (nc - RD + NC) % NC;
    int n = 2 * (NQ - 1) + P(c0) + (c0 == NC - 1 ? NT1S : 0) + NW1 + NW3 + (nc == NC - 1 ? AF : 0);
    for (int k = 1; k < RD; ++k) n += tot((c0 + k) % NC);
    return n;
  }
  static_assert(CH == 1 || CH == 2, "a wave, or a pair of waves, per 32 tile rows");
  static_assert(C / N1 == 4 || C / N1 == 2, "t1's row offsets are derived from y's by a shift");
  static_assert(!STAGE || (PT2 * NC == 2 * NT2 && PT2 >= 1 && NC >= 4), "the t2 pieces of the next tile ride on the first half of the chunks");
  static_assert(NC % RD == 0, "the chunk loop is unrolled by the residual depth");
  static_assert(kBytes <= 160 * 1024, "LDS budget");
};

#ifndef TSM_C31_X
#define TSM_C31_X 0    // timing experiments only (wrong results): 1 no GEMM1, 2 no GEMM2, 4 no residual loads, 8 no y stores, 16 no weight DMA, 32 no chunk epilogue ALU/LDS
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt_imm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wait_vmcnt_any(int n) {
#define TSM_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    TSM_VMCNT_CASE(0) TSM_VMCNT_CASE(1) TSM_VMCNT_CASE(2) TSM_VMCNT_CASE(3) TSM_VMCNT_CASE(4) TSM_VMCNT_CASE(5)
    TSM_VMCNT_CASE(6) TSM_VMCNT_CASE(7) TSM_VMCNT_CASE(8) TSM_VMCNT_CASE(9) TSM_VMCNT_CASE(10) TSM_VMCNT_CASE(11)
    TSM_VMCNT_CASE(12) TSM_VMCNT_CASE(13) TSM_VMCNT_CASE(14) TSM_VMCNT_CASE(15) TSM_VMCNT_CASE(16) TSM_VMCNT_CASE(17)
    TSM_VMCNT_CASE(18) TSM_VMCNT_CASE(19) TSM_VMCNT_CASE(20) TSM_VMCNT_CASE(21) TSM_VMCNT_CASE(22) TSM_VMCNT_CASE(23)
    TSM_VMCNT_CASE(24) TSM_VMCNT_CASE(25) TSM_VMCNT_CASE(26) TSM_VMCNT_CASE(27) TSM_VMCNT_CASE(28) TSM_VMCNT_CASE(29)
    TSM_VMCNT_CASE(30) TSM_VMCNT_CASE(31) TSM_VMCNT_CASE(32) TSM_VMCNT_CASE(33) TSM_VMCNT_CASE(34) TSM_VMCNT_CASE(35)
    TSM_VMCNT_CASE(36) TSM_VMCNT_CASE(37) TSM_VMCNT_CASE(38) TSM_VMCNT_CASE(39) TSM_VMCNT_CASE(40) TSM_VMCNT_CASE(41)
    TSM_VMCNT_CASE(42) TSM_VMCNT_CASE(43) TSM_VMCNT_CASE(44) TSM_VMCNT_CASE(45) TSM_VMCNT_CASE(46) TSM_VMCNT_CASE(47)
    TSM_VMCNT_CASE(48)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef TSM_VMCNT_CASE
}

template <int K3, int C, int N1, int CH>
__global__ void __launch_bounds__(512, 2) conv31_fused_kernel(const Conv31Params p) {
  typedef C31<K3, C, N1, CH> L;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave / CH, wh = wave % CH;               // row group (32 tile rows) and this wave's share of its columns
  const int half = lane >> 5, l31 = lane & 31, c8 = lane & 7, r8l = lane >> 3;
  const int T = p.T, HW = p.HW;
  const int lpx = p.log_px, PX = 1 << lpx;               // pixels of a tile: rows / T
  const int tpc = (HW + PX - 1) >> lpx;                   // tiles per clip
  const int ntiles = p.n_clips * tpc, nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int my = (ntiles - bid + nwg - 1) / nwg;          // tiles of this workgroup (>= 1: the grid never exceeds the tiles)

  float *bias3_l = reinterpret_cast<float *>(lds + L::kBias3), *bias1_l = reinterpret_cast<float *>(lds + L::kBias1);
  for (int i = tid; i < C; i += L::NT) bias3_l[i] = p.bias3[i];
  for (int i = tid; i < N1; i += L::NT) bias1_l[i] = p.bias1[i];
  if (tid < 32) reinterpret_cast<unsigned *>(lds + L::kZero)[tid] = 0u;

  // ---- per-lane constants -------------------------------------------------------------------------------------
  // epilogue step q of this wave: row 8 q + r8l of its share of the row group (CH = 1: all 32 rows, CH = 2: 16), channels
  // 8 c8 .. 8 c8 + 7 of the chunk
  unsigned evoff[L::NQ], epx[L::NQ], yw[L::NQ];
#pragma unroll
  for (int q = 0; q < L::NQ; ++q) {
    const int row = 32 * rg + 8 * L::NQ * wh + 8 * q + r8l;
    const int t = row >> lpx, px = row & (PX - 1);
    evoff[q] = (unsigned)((t * HW + px) * (C * 2) + c8 * 16);          // byte offset in the clip's [T*HW][C] block (+ p0 * C * 2)
    epx[q] = (unsigned)px;
    yw[q] = (unsigned)(L::kY + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
  }
  // GEMM2's A fragments: tile row r = 32 rg + l31 as it stands, or rows r + PX / r - PX (frames t + 1 / t - 1), or zeros
  unsigned ybase[3], yflip[3];
  const int arow = 32 * rg + l31;
  {
    const int t = arow >> lpx, rp = arow + PX, rm = arow - PX;
    ybase[0] = (unsigned)(L::kY + arow * 128);   yflip[0] = (unsigned)((arow >> 1) & 7);
    ybase[1] = t + 1 < T ? (unsigned)(L::kY + rp * 128) : (unsigned)L::kZero;   yflip[1] = t + 1 < T ? (unsigned)((rp >> 1) & 7) : 0u;
    ybase[2] = t > 0 ? (unsigned)(L::kY + rm * 128) : (unsigned)L::kZero;       yflip[2] = t > 0 ? (unsigned)((rm >> 1) & 7) : 0u;
  }
  // fragment reads: row * row bytes + ((2 g + half) ^ flip) * 16 = (base ^ (g << 5)) with base = row * row bytes +
  // ((half ^ flip & 1) << 4) + ((flip >> 1) << 5) -- the row starts are multiples of 128 / 256 / 512, so bits 5.. of the
  // base hold nothing but the flip
  const unsigned w1flip = (unsigned)((l31 >> 1) & 7), rflip = (unsigned)(l31 & 15);
  const unsigned w1a = (unsigned)(L::kW1 + (wh * (N1 / CH) + l31) * 128) + (((unsigned)half ^ (w1flip & 1u)) << 4) + ((w1flip >> 1) << 5);
  const unsigned w3a = (unsigned)(L::kW3 + (wh * (64 / CH) + l31) * L::RB3) + (((unsigned)half ^ (rflip & 1u)) << 4) + ((rflip >> 1) << 5);
  static_assert(L::kW3 % 512 == 0 && L::kW1 % 128 == 0 && L::kY % 128 == 0 && L::kZero % 128 == 0 && L::kT2 % 512 == 0,
                "XOR addressing of the fragment reads");
  const unsigned t2rd = (unsigned)(L::kT2 + arow * L::RB3);
  // direct A-fragment loads (CH = 2): 16 bytes of row `arow`, k16 group g at + 32 g
  const unsigned afoff = (unsigned)(((arow >> lpx) * HW + (arow & (PX - 1))) * L::RB3 + half * 16), afpx = (unsigned)(arow & (PX - 1));
  // DMA sources (the 16-byte chunk a lane fetches is swizzled on the SOURCE side; the LDS side is linear per piece)
  unsigned w3off[L::NW3], w1off[L::NW1];
#pragma unroll
  for (int i = 0; i < L::NW3; ++i) {
    const int row = (wave * L::NW3 + i) * L::RPP3 + lane / L::LPR3, slot = lane % L::LPR3;
    w3off[i] = (unsigned)(row * L::RB3 + ((slot ^ (row & 15)) << 4));
  }
#pragma unroll
  for (int i = 0; i < L::NW1; ++i) {
    const int row = (wave * L::NW1 + i) * 8 + (lane >> 3), slot = lane & 7;
    w1off[i] = (unsigned)(row * (C * 2) + ((slot ^ ((row >> 1) & 7)) << 4));
  }
  const __amdgpu_buffer_rsrc_t rsrcW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w3), 0, C * K3 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, N1 * C * 2, 0x00020000);
  const size_t clip_rows = (size_t)T * HW;
  // (plain ints, not L:: constants, inside the scalar-offset arguments of the buffer builtins below: with a template-dependent
  //  constant expression there the HOST pass of hipcc silently drops the kernel's stub from the object -- no diagnostic)
  const int rb3 = L::RB3, cb = C * 2, n1b = N1 * 2;

  auto tile_of = [&](int s, int *clip, int *p0) {
    int tile = bid + s * nwg;
    if (tile >= ntiles) tile = ntiles - 1;                // (s == my: staged dead, kept in range for the arithmetic)
    if (p.reverse) tile = ntiles - 1 - tile;
    *clip = tile / tpc;
    *p0 = (tile - *clip * tpc) << lpx;
  };
  auto issue_w3 = [&](int nc, unsigned dead) {
#pragma unroll
    for (int i = 0; i < ((TSM_C31_X & 16) ? 0 : L::NW3); ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW3, (lds_void *)(lds + L::kW3 + (wave * L::NW3 + i) * 1024), 16,
                                               (int)(w3off[i] | dead), nc * 64 * rb3, 0, 0);
  };
  auto issue_w1_piece = [&](int nc, int i) {
    if (!(TSM_C31_X & 16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW1, (lds_void *)(lds + L::kW1 + (wave * L::NW1 + i) * 1024), 16,
                                             (int)w1off[i], nc * 128, 0, 0);
  };
  // STAGE: pieces i0 .. i0 + n - 1 of the t2 tile (clip, p0) into the staging buffer
  auto issue_t2 = [&](int clip, int p0, int i0, int n, unsigned dead) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.t2) + (size_t)clip * clip_rows * L::RB3), 0, (int)(clip_rows * L::RB3), 0x00020000);
    for (int i = i0; i < i0 + n; ++i) {
      const int row = 32 * wave + i * L::RPP3 + lane / L::LPR3, slot = lane % L::LPR3;
      const int t = row >> lpx, px = row & (PX - 1);
      const unsigned off = (unsigned)((t * HW + px) * L::RB3 + ((slot ^ (row & 15)) << 4));
      const unsigned inv = p0 + px < HW ? 0u : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(lds + L::kT2 + (wave * L::NT2 + i) * 1024), 16, (int)(off | inv | dead),
                                               p0 * rb3, 0, 0);
    }
  };
  u32x4 afr[L::KT1];                                      // GEMM1's A operand: this wave's 32 rows of t2, held for a whole tile
  auto load_afr = [&](int clip, int p0, unsigned dead) {  // !STAGE: straight from global memory
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.t2) + (size_t)clip * clip_rows * L::RB3), 0, (int)(clip_rows * L::RB3), 0x00020000);
    const unsigned inv = (unsigned)p0 + afpx < (unsigned)HW ? 0u : kInvalid;
#pragma unroll
    for (int g = 0; g < L::KT1; ++g)
      afr[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(afoff | inv | dead), p0 * rb3 + g * 32, 0);
  };
  // the residual of a chunk's epilogue steps, requested RD chunks ahead (register set = chunk index mod RD)
  u32x4 rres[L::RD * L::NQ];
  auto load_res = [&](int slot, int q, int clip, int p0, int nc, unsigned dead) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.res) + (size_t)clip * clip_rows * (C * 2)), 0, (int)(clip_rows * (C * 2)), 0x00020000);
    const unsigned inv = (unsigned)p0 + epx[q] < (unsigned)HW ? 0u : kInvalid;
    if (!(TSM_C31_X & 4)) rres[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(evoff[q] | inv | dead), p0 * cb + nc * 128, 0);
    else rres[slot] = u32x4{(unsigned)slot, 0u, 0u, 0u};
  };

  // ---- prologue: the first tile's t2, the first chunk of W3, the first chunk's residual ----------------------------
  int clip, p0, nclip, np0;
  tile_of(0, &clip, &p0);
  if constexpr (L::STAGE) issue_t2(clip, p0, 0, L::NT2, 0u);
  else load_afr(clip, p0, 0u);
  issue_w3(0, 0u);
#pragma unroll
  for (int r = 0; r < L::RD; ++r)
#pragma unroll
    for (int q = 0; q < L::NQ; ++q) load_res(r * L::NQ + q, q, clip, p0, r, 0u);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // fp32 slab of the chunk epilogue: STAGE this wave's own [8][68]; else the pair's [32][68] (its first 2176 bytes per wave
  // double as the wave's private sub-slab in the t1 epilogue)
  float *Cs = reinterpret_cast<float *>(lds + L::kSlab + (L::STAGE ? wave * 2176 : rg * 8704));
  float *Cw = L::STAGE ? Cs : Cs + wh * 544;
  for (int s = 0; s < my; ++s) {
    const unsigned next_dead = s + 1 < my ? 0u : kInvalid;
    tile_of(s + 1, &nclip, &np0);
    if constexpr (L::STAGE) {
      // (every wave's pieces landed behind counted waits and barriers of the previous tile's second half -- or the prologue)
#pragma unroll
      for (int g = 0; g < L::KT1; ++g)
        afr[g] = *reinterpret_cast<const u32x4 *>(lds + t2rd + (((2 * g + half) ^ rflip) << 4));
    }
    if constexpr (!L::STAGE) {
      // The A fragments were requested in the previous tile's last chunk.  Consuming them HERE, once per tile, makes the
      // compiler place its wait for these register loads here too: left to the first MFMA inside the chunk loop, its
      // loop-carried analysis put a conservative vmcnt(3) in front of EVERY chunk's GEMM1, which also waited for the
      // residual loads of the chunk -- the stream this kernel lives on.
#pragma unroll
      for (int g = 0; g < L::KT1; ++g) asm volatile("" ::"v"(afr[g]));
    }
    f32x16 acc2[L::NTL2];
#pragma unroll
    for (int j = 0; j < L::NTL2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the staging buffer is refilled from chunk 0's epilogue on: behind barrier A)
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.y) + (size_t)clip * clip_rows * (C * 2), 0, (int)(clip_rows * (C * 2)), 0x00020000);
    unsigned einv[L::NQ];
#pragma unroll
    for (int q = 0; q < L::NQ; ++q) einv[q] = (unsigned)p0 + epx[q] < (unsigned)HW ? 0u : kInvalid;

    for (int nc0 = 0; nc0 < L::NC; nc0 += L::RD) {
#pragma unroll
     for (int rset = 0; rset < L::RD; ++rset) {
      const int nc = nc0 + rset;
      const bool last = nc + 1 == L::NC;
      const int pnow = nc < L::NC / 2 ? L::PT2 : 0;                        // t2 pieces of the next tile issued in this chunk
      // W3's chunk nc has landed (this wave's pieces)
      if (nc == 0) wait_vmcnt_imm<L::wait_w3(0)>();
      else if (nc - 1 < L::NC / 2) wait_vmcnt_imm<L::wait_w3(1)>();
      else wait_vmcnt_imm<L::wait_w3(L::NC - 1)>();
      __builtin_amdgcn_s_barrier();                                        // A: ... everybody's; GEMM2 of chunk nc - 1 is over
#pragma unroll
      for (int i = 0; i < L::NW1; ++i) issue_w1_piece(nc, i);
      // ---- GEMM1: y[rows of this row group][this wave's channels of the chunk] ----
      f32x16 acc1[L::NTL1];
#pragma unroll
      for (int j = 0; j < L::NTL1; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
      {
        // B fragments through a ring of four, read four MFMAs ahead (left to the compiler the loop was read-two / wait /
        // multiply-two).  Fragment (g, j) sits at (ba ^ (g << 5)) + j * 32 rows: the row's XOR swizzle as ONE v_xor per read
        // instead of KT1 precomputed address registers.
        unsigned ba = w3a;
        asm volatile("" : "+v"(ba));                       // (keeps the KT1 addresses from being hoisted out of the chunk loop)
        constexpr int NM = L::KT1 * L::NTL1, D = 4;
        u32x4 ring[D];
#pragma unroll
        for (int m = 0; m < D; ++m)
          ring[m] = *reinterpret_cast<const u32x4 *>(lds + (ba ^ (unsigned)((m / L::NTL1) << 5)) + (m % L::NTL1) * 32 * L::RB3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < ((TSM_C31_X & 1) ? 0 : NM); ++m) {
          const u32x4 b = ring[m % D];
          if (m + D < NM)
            ring[m % D] = *reinterpret_cast<const u32x4 *>(lds + (ba ^ (unsigned)(((m + D) / L::NTL1) << 5)) + ((m + D) % L::NTL1) * 32 * L::RB3);
          acc1[m % L::NTL1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[m / L::NTL1]), __builtin_bit_cast(bf16x8, b),
                                                                      acc1[m % L::NTL1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks the reads back next to their MFMA)
        }
      }
      if constexpr (!L::STAGE) {    // the pair's slab: this wave's 32 columns of all 32 rows, complete behind barrier B
#pragma unroll
        for (int e = 0; e < 16; ++e) Cs[((e & 3) + 8 * (e >> 2) + 4 * half) * 68 + wh * 32 + l31] = acc1[0][e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                        // B: every wave has read W3's chunk
      issue_w3(last ? 0 : nc + 1, last ? next_dead : 0u);
      if constexpr (!L::STAGE) {
        if (last) load_afr(nclip, np0, next_dead);                         // GEMM1 of this tile is over: the next tile's A fragments
      }
      // ---- epilogue of the chunk: + bias3, + residual, ReLU, bf16 -> y (global) and the LDS tile ----
      const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(bias3_l + nc * 64 + c8 * 8);
      const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(bias3_l + nc * 64 + c8 * 8 + 4);
#pragma unroll
      for (int q = 0; q < ((TSM_C31_X & 32) ? 0 : L::NQ); ++q) {
        f32x4 c0, c1;
        if constexpr (L::STAGE) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(4 * half + r) * 68 + j * 32 + l31] = acc1[j][4 * q + r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (same wave wrote it: no barrier needed)
          c0 = *reinterpret_cast<const f32x4 *>(Cs + r8l * 68 + c8 * 8);
          c1 = *reinterpret_cast<const f32x4 *>(Cs + r8l * 68 + c8 * 8 + 4);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // read before the next sub-slab overwrites it
        } else {
          c0 = *reinterpret_cast<const f32x4 *>(Cs + (16 * wh + 8 * q + r8l) * 68 + c8 * 8);
          c1 = *reinterpret_cast<const f32x4 *>(Cs + (16 * wh + 8 * q + r8l) * 68 + c8 * 8 + 4);
        }
        // this step's residual (requested one chunk ago)
        // this step's residual (requested RD chunks ago); the values differ where the window crosses the tile's end
        static_assert(L::RD <= 2 && (L::STAGE ? L::RD == 1 : L::wait_res(2) == L::wait_res(L::NC - 2)), "the cases below");
        if (nc == 0) wait_vmcnt_imm<L::wait_res(0)>();
        else if (nc == 1) wait_vmcnt_imm<L::wait_res(1)>();
        else if (last) wait_vmcnt_imm<L::wait_res(L::NC - 1)>();
        else if (L::STAGE && nc - 1 < L::NC / 2) wait_vmcnt_imm<L::wait_res(2)>();
        else wait_vmcnt_imm<L::wait_res(L::NC - 2)>();
        // (pins the residual's first use behind the counted wait: the scheduler otherwise hoists its bf16 -> fp32 unpacking into
        //  GEMM1's MFMA shadow, and the compiler's own wait for these registers then sits in front of GEMM1)
        asm volatile("" : "+v"(rres[rset * L::NQ + q]));
        float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                      c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += split_elem(rres[rset * L::NQ + q], e);
        u32x4 o;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) o[w2] = pack_bf16(fmaxf(v[2 * w2], 0.f), fmaxf(v[2 * w2 + 1], 0.f));
        if (!(TSM_C31_X & 8)) __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)(evoff[q] | einv[q]), p0 * cb + nc * 128, TSM_AUX_C31);
        *reinterpret_cast<u32x4 *>(lds + yw[q]) = o;
        if (nc + L::RD < L::NC) load_res(rset * L::NQ + q, q, clip, p0, nc + L::RD, 0u);
        else load_res(rset * L::NQ + q, q, nclip, np0, nc + L::RD - L::NC, next_dead);
      }
      if constexpr (L::STAGE) {
        if (pnow > 0) issue_t2(nclip, np0, L::PT2 * nc, L::PT2, next_dead);
      }
      // W1's chunk nc has landed (this wave's pieces)
      if (last) wait_vmcnt_imm<L::wait_w1(L::NC - 1)>();
      else if (nc < L::NC / 2) wait_vmcnt_imm<L::wait_w1(0)>();
      else wait_vmcnt_imm<L::wait_w1(L::NC - 2)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // ... and this wave's rows of the LDS tile are written
      __builtin_amdgcn_s_barrier();                                        // C
      // ---- GEMM2: t1 += shift(y chunk) * W1[:, chunk]^T ----
      const int c0ch = nc * 64;
      const int sel = c0ch < p.fold ? 1 : (c0ch < 2 * p.fold ? 2 : 0);     // wave-uniform: frames t + 1 / t - 1 / t
      const unsigned yb = sel == 1 ? ybase[1] : sel == 2 ? ybase[2] : ybase[0];
      const unsigned yf = sel == 1 ? yflip[1] : sel == 2 ? yflip[2] : yflip[0];
      {
        // the four A fragments up front, the B fragments through a ring of four (see GEMM1)
        unsigned ya = yb + (((unsigned)half ^ (yf & 1u)) << 4) + ((yf >> 1) << 5), bb = w1a;
        asm volatile("" : "+v"(ya), "+v"(bb));
        u32x4 a4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) a4[g] = *reinterpret_cast<const u32x4 *>(lds + (ya ^ (unsigned)(g << 5)));
        constexpr int NM = 4 * L::NTL2, D = 4;
        u32x4 ring[D];
#pragma unroll
        for (int m = 0; m < D; ++m)
          ring[m] = *reinterpret_cast<const u32x4 *>(lds + (bb ^ (unsigned)((m / L::NTL2) << 5)) + (m % L::NTL2) * 32 * 128);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < ((TSM_C31_X & 2) ? 0 : NM); ++m) {
          const u32x4 b = ring[m % D];
          if (m + D < NM)
            ring[m % D] = *reinterpret_cast<const u32x4 *>(lds + (bb ^ (unsigned)(((m + D) / L::NTL2) << 5)) + ((m + D) % L::NTL2) * 32 * 128);
          acc2[m % L::NTL2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4[m / L::NTL2]), __builtin_bit_cast(bf16x8, b),
                                                                      acc2[m % L::NTL2], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
     }
    }
    // ---- t1 of the tile: relu(acc2 + bias1) -> bf16, whole 128-byte row segments (this wave's N1 / CH columns) ----
    const __amdgpu_buffer_rsrc_t rsrcT1 = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.t1) + (size_t)clip * clip_rows * (N1 * 2), 0, (int)(clip_rows * (N1 * 2)), 0x00020000);
#pragma unroll
    for (int jh = 0; jh < L::NTL2 / 2; ++jh) {
      const int col0 = wh * (N1 / CH) + jh * 64;
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias1_l + col0 + c8 * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias1_l + col0 + c8 * 8 + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cw[(4 * half + r) * 68 + j * 32 + l31] = acc2[2 * jh + j][4 * q + r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cw + r8l * 68 + c8 * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cw + r8l * 68 + c8 * 8 + 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float v[8] = {c0[0] + b0[0], c0[1] + b0[1], c0[2] + b0[2], c0[3] + b0[3],
                            c1[0] + b1[0], c1[1] + b1[1], c1[2] + b1[2], c1[3] + b1[3]};
        u32x4 o;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) o[w2] = pack_bf16(fmaxf(v[2 * w2], 0.f), fmaxf(v[2 * w2 + 1], 0.f));
        // row 32 rg + 8 q + r8l of the tile (all 32 rows of the group: the pair splits t1 by COLUMNS)
        const int row = 32 * rg + 8 * q + r8l, t = row >> lpx, px = row & (PX - 1);
        const unsigned off = (unsigned)((t * HW + px) * (N1 * 2) + c8 * 16), inv = p0 + px < HW ? 0u : kInvalid;
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcT1, (int)(off | inv), p0 * n1b + col0 * 2, TSM_AUX_C31);
      }
    }
    clip = nclip;
    p0 = np0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the dead tail stages (zeros) land before the workgroup leaves its LDS
}

// ---------------------------------------------------------------------------------------------
// conv31_pc_kernel<K3, C, N1>: the same fusion on the 128-row tiles (N1 = 256: layer2.3 -> layer3.0, layer3.k -> layer3.k+1)
// with the eight waves SPECIALISED instead of paired (round 5).
//
// Why: in conv31_fused_kernel<.., CH = 2> a chunk is three barrier-separated phases -- GEMM1 (LDS port + matrix pipe), the
// chunk epilogue (vector ALU + memory queue), GEMM2 (LDS port + matrix pipe) -- and all eight waves sit in the SAME phase,
// so nothing overlaps anything: profiles/r04_c31_work_removal.txt adds the phases up to the launch (81 + 78 + 88 + 110 of
// 393 us), 0.28 of the matrix pipe and 3.6 TB/s, neither roof.  Here the two waves of every SIMD have different jobs:
//   * waves 0-3, the PRODUCERS (32 tile rows each): GEMM1 of chunk k from their register-resident t2 fragments and the
//     chunk of W3 in LDS (both N-tiles: 2 K3 / 16 MFMAs), then the chunk epilogue through a wave-private [8][68] fp32
//     sub-slab -- + bias3, + residual, ReLU, bf16 -> y (global) and the [128][64] LDS tile of chunk k; they own the
//     activation stream (t2 fragments of the next tile, residual RD chunks ahead, y stores: plain register loads / stores);
//   * waves 4-7, the CONSUMERS (the SIMD partners; 32 tile rows each): GEMM2 of chunk k - 1 -- t1 += shift(y chunk) * W1 chunk,
//     32 x 256 accumulators per wave, 32 MFMAs -- while the producers are on chunk k; at a tile's end the t1 epilogue; they
//     own the WEIGHT stream: every LDS-DMA piece of W3 / W1 is theirs, so no wave mixes the two kinds of vector-memory
//     operation and each role's counted waits see only its own queue.
// On a SIMD the producer's epilogue (vector ALU) runs beside the consumer's MFMAs, and the two GEMMs share the matrix pipe
// back to back instead of taking turns with an idle pipe in between.  The LDS tile and the W1 chunk are double-buffered
// (chunk k is written / fetched while chunk k - 1 is multiplied); W3's chunk is single: its refill is issued behind barrier B,
// when the producers have read it, and lands under the epilogue.  TWO barriers per chunk slot:
//     A | P: GEMM1(k)                      Q: DMA W1(k) -> buffer k & 1; first QSPLIT MFMAs of GEMM2(k - 1)
//     B | P: [last chunk: next tile's t2]  Q: DMA W3(k + 1); the rest of GEMM2(k - 1); [k - 1 last: the tile's t1 epilogue];
//       |    epilogue(k) -> y, LDS tile k & 1     counted wait: its DMA pieces have landed (the t1 stores are younger)
// The roles run in two separate loops with the same barrier sequence (no control-flow join inside: the compiler's own wait
// insertion never sees the other role's pending operations).  Same products in the same order per accumulator, same two
// epilogues as conv31_fused_kernel -> bit-identical to it and to the two launches it replaces.
// ---------------------------------------------------------------------------------------------
#ifndef TSM_C31P_RD
#define TSM_C31P_RD 2      // chunks the residual is requested ahead
#endif
#ifndef TSM_C31P_QSPLIT
#define TSM_C31P_QSPLIT 0  // GEMM2 MFMAs a consumer issues in front of barrier B (measured: 0 / 8 / 16 within 2 %, 0 best)
#endif
#ifndef TSM_C31P_SPREAD
#define TSM_C31P_SPREAD 0  // 1: the consumers' weight pieces ride one by one behind GEMM2's MFMAs instead of in two bursts (measured with
                           // QSPLIT 16: 366 -> 391 us per layer3 site -- the producers wait for W3 at the next A, the bursts land it sooner)
#endif
template <int K3, int C, int N1> struct C31P {
  static constexpr int M = 128, NT = 512;
  static constexpr int KT1 = K3 / 16;              // k16 steps of GEMM1
  static constexpr int NC = C / 64;                // chunks of the block's channels
  static constexpr int NQ = 4;                     // 8-row epilogue steps per producer and chunk
  static constexpr int RD = TSM_C31P_RD;
  static constexpr int RB3 = K3 * 2, LPR3 = RB3 / 16, RPP3 = 1024 / RB3;
  static constexpr int NW3 = 64 * RB3 / 1024 / 4;  // DMA pieces per CONSUMER wave: a chunk of W3 (64 rows)
  static constexpr int NW1 = N1 * 128 / 1024 / 4;  // ... a chunk of W1 (N1 rows x 64 channels)
  static constexpr int NTL2 = N1 / 32;             // GEMM2 N-tiles per consumer
  static constexpr int NT1S = 4 * (N1 / 64);       // t1 stores per consumer and tile
  static constexpr int AF = KT1;                   // register loads of the next tile's A fragments, in a tile's last chunk
  static constexpr int QSPLIT = TSM_C31P_QSPLIT;
  static constexpr int kW3 = 0;
  static constexpr int kW1 = kW3 + 64 * RB3;       // two buffers
  static constexpr int kY = kW1 + 2 * N1 * 128;    // two buffers
  static constexpr int kSlab = kY + 2 * M * 128;   // wave-private fp32 sub-slabs: [16][68] per producer, [8][68] per consumer
  static constexpr int kBias3 = kSlab + 4 * 4352 + 4 * 2176;
  static constexpr int kBias1 = kBias3 + C * 4;
  static constexpr int kZero = kBias1 + N1 * 4;
  static constexpr int kBytes = kZero + 128;
  // Producer's vector-memory operations in issue order, per chunk c: [c last: AF loads] (store y, load the residual of chunk c + RD) x NQ.
  // The residual of a step of chunk nc was issued at the same step RD chunks earlier: younger than it are the rest of that
  // chunk's steps, RD - 1 whole chunks, and this chunk up to the step -- the same number for every step:
  static constexpr int wait_res(int nc) {
    int n = 2 * (NQ - 1) + 2 * NQ * (RD - 1);
    for (int k = 1; k <= RD; ++k) n += ((nc - RD + k + NC) % NC == NC - 1) ? AF : 0;   // a tile's last chunk in (nc - RD, nc]
    return n;
  }
  static_assert(NC % RD == 0, "the chunk loop is unrolled by the residual depth");
  static_assert(kW3 % 512 == 0 && kW1 % 128 == 0 && kY % 128 == 0 && kZero % 128 == 0, "XOR addressing of the fragment reads");
  static_assert(kBytes <= 160 * 1024, "LDS budget");
  static_assert(QSPLIT >= 0 && QSPLIT <= 4 * NTL2, "GEMM2 is 4 NTL2 MFMAs");
};

#ifndef TSM_C31P_STAMP
#define TSM_C31P_STAMP 0   // diagnostic builds only: per-phase cycle sums of workgroup 0 (s_memtime), printed at the kernel's end
#endif
#if TSM_C31P_STAMP
#define C31P_STAMP(i)                                       \
  do {                                                      \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += _t - stamp_last;                        \
    stamp_last = _t;                                        \
  } while (0)
#else
#define C31P_STAMP(i) do {} while (0)
#endif
template <int K3, int C, int N1>
__global__ void __launch_bounds__(512, 2) conv31_pc_kernel(const Conv31Params p) {
  typedef C31P<K3, C, N1> L;
#if TSM_C31P_STAMP
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int rg = wave & 3;                                // row group: 32 tile rows
  const int half = lane >> 5, l31 = lane & 31, c8 = lane & 7, r8l = lane >> 3;
  const int T = p.T, HW = p.HW;
  const int lpx = p.log_px, PX = 1 << lpx;
  const int tpc = (HW + PX - 1) >> lpx;                   // tiles per clip
  const int ntiles = p.n_clips * tpc, nwg = (int)gridDim.x, bid = (int)blockIdx.x;
  const int my = (ntiles - bid + nwg - 1) / nwg;          // tiles of this workgroup (>= 1: the grid never exceeds the tiles)
  const int nslots = my * L::NC;                          // chunk slots; one more drains the consumers

  float *bias3_l = reinterpret_cast<float *>(lds + L::kBias3), *bias1_l = reinterpret_cast<float *>(lds + L::kBias1);
  for (int i = tid; i < C; i += L::NT) bias3_l[i] = p.bias3[i];
  for (int i = tid; i < N1; i += L::NT) bias1_l[i] = p.bias1[i];
  if (tid < 32) reinterpret_cast<unsigned *>(lds + L::kZero)[tid] = 0u;

  const size_t clip_rows = (size_t)T * HW;
  // (plain ints, not L:: constants, inside the scalar-offset arguments of the buffer builtins: see conv31_fused_kernel)
  const int rb3 = L::RB3, cb = C * 2, n1b = N1 * 2;
  auto tile_of = [&](int s, int *clip, int *p0) {
    int tile = bid + s * nwg;
    if (tile >= ntiles) tile = ntiles - 1;                // (s == my: dead, kept in range for the arithmetic)
    if (p.reverse) tile = ntiles - 1 - tile;
    *clip = tile / tpc;
    *p0 = (tile - *clip * tpc) << lpx;
  };
  // this wave's fp32 sub-slab: 16 rows for a producer (two epilogue steps per LDS round trip), 8 for a consumer (the t1 epilogue)
  float *Cw = reinterpret_cast<float *>(lds + L::kSlab + (producer ? wave * 4352 : 4 * 4352 + (wave - 4) * 2176));

  if (producer) {
    // ================================ PRODUCERS: GEMM1 + the chunk epilogue ================================
    // epilogue step q: row 32 rg + 8 q + r8l of the tile, channels 8 c8 .. 8 c8 + 7 of the chunk
    unsigned evoff[L::NQ], epx[L::NQ], yw[L::NQ];
#pragma unroll
    for (int q = 0; q < L::NQ; ++q) {
      const int row = 32 * rg + 8 * q + r8l;
      const int t = row >> lpx, px = row & (PX - 1);
      evoff[q] = (unsigned)((t * HW + px) * (C * 2) + c8 * 16);          // byte offset in the clip's [T*HW][C] block (+ p0 * C * 2)
      epx[q] = (unsigned)px;
      yw[q] = (unsigned)(L::kY + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
    }
    const int arow = 32 * rg + l31;
    const unsigned rflip = (unsigned)(l31 & 15);
    const unsigned w3a = (unsigned)(L::kW3 + l31 * L::RB3) + (((unsigned)half ^ (rflip & 1u)) << 4) + ((rflip >> 1) << 5);
    const unsigned afoff = (unsigned)(((arow >> lpx) * HW + (arow & (PX - 1))) * L::RB3 + half * 16), afpx = (unsigned)(arow & (PX - 1));
    u32x4 afr[L::KT1];                                    // GEMM1's A operand: this wave's 32 rows of t2, held for a whole tile
    // (inline asm like the residual loads below, for the same reason: with the builtin, hipcc cannot tell that the loads of a
    //  tile's LAST chunk are consumed by the next tile only, and puts a wait ladder for them in front of every other chunk's GEMM1)
    auto load_afr = [&](int clip, int p0, unsigned dead) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.t2) + (size_t)clip * clip_rows * L::RB3), 0, (int)(clip_rows * L::RB3), 0x00020000);
      const unsigned inv = (unsigned)p0 + afpx < (unsigned)HW ? 0u : kInvalid;
      const unsigned voff = afoff | inv | dead;
      const int soff = __builtin_amdgcn_readfirstlane(p0 * rb3);
      asm volatile("s_nop 4" ::: "memory");
#pragma unroll
      for (int g = 0; g < L::KT1; ++g)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(afr[g]) : "v"(voff), "s"(rsrc), "s"(soff), "n"(g * 32) : "memory");
    };
    // The residual loads are issued from inline asm: hipcc's own wait insertion then does not see them.  Left to the compiler
    // (round 4's kernel), every use of a residual register got a second, much tighter wait behind the counted one -- vmcnt(11)
    // behind vmcnt(46), vmcnt(7) behind vmcnt(30): its loop-carried bookkeeping allows only the previous chunk's operations to
    // stay in flight, i.e. the RD-deep request stream was cut to one chunk.  Form (ii) of cdna_hip_programming.md 5.7: "=v"
    // loads, then the counted wait names the destination "+v" in front of its first consumer; the ISA is audited for compiler
    // moves of these registers between load and wait (tests/test_code_objects.py).  The builtin stores and t2 loads stay
    // visible to the compiler: what it does not count can only make ITS waits tighter than needed, never looser.
    u32x4 rres[L::RD * L::NQ];                            // residual register sets: chunk index mod RD
    auto load_res = [&](int slot, int q, int clip, int p0, int nc, unsigned dead) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(reinterpret_cast<const char *>(p.res) + (size_t)clip * clip_rows * (C * 2)), 0, (int)(clip_rows * (C * 2)), 0x00020000);
      const unsigned inv = (unsigned)p0 + epx[q] < (unsigned)HW ? 0u : kInvalid;
      const unsigned voff = evoff[q] | inv | dead;
      const int soff = __builtin_amdgcn_readfirstlane(p0 * cb + nc * 128);
      // (s_nop 4: the scalar operands may be fresh from a v_readfirstlane; nothing inside an asm statement is padded)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rres[slot]) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    };

    int clip, p0, nclip, np0;
    tile_of(0, &clip, &p0);
    load_afr(clip, p0, 0u);
#pragma unroll
    for (int r = 0; r < L::RD; ++r)
#pragma unroll
      for (int q = 0; q < L::NQ; ++q) load_res(r * L::NQ + q, q, clip, p0, r, 0u);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                                          // P0: biases, the zero row, W3's first chunk (consumers)

    unsigned ybuf = 0u;                                                    // LDS tile buffer of this slot: (slot & 1) * M * 128
    for (int s = 0; s < my; ++s) {
      const unsigned next_dead = s + 1 < my ? 0u : kInvalid;
      tile_of(s + 1, &nclip, &np0);
      // the tile's A fragments (requested in the previous tile's last chunk, in front of its 2 NQ epilogue operations -- or in the
      // prologue, which waited for everything: the count is then merely generous)
      static_assert(L::KT1 == 8 || L::KT1 == 16, "operand lists below");
      if constexpr (L::KT1 == 16)
        asm volatile("s_waitcnt vmcnt(%16)"
                     : "+v"(afr[0]), "+v"(afr[1]), "+v"(afr[2]), "+v"(afr[3]), "+v"(afr[4]), "+v"(afr[5]), "+v"(afr[6]), "+v"(afr[7]),
                       "+v"(afr[8 % L::KT1]), "+v"(afr[9 % L::KT1]), "+v"(afr[10 % L::KT1]), "+v"(afr[11 % L::KT1]), "+v"(afr[12 % L::KT1]),
                       "+v"(afr[13 % L::KT1]), "+v"(afr[14 % L::KT1]), "+v"(afr[15 % L::KT1])
                     : "n"(2 * L::NQ) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(afr[0]), "+v"(afr[1]), "+v"(afr[2]), "+v"(afr[3]), "+v"(afr[4]), "+v"(afr[5]), "+v"(afr[6]), "+v"(afr[7])
                     : "n"(2 * L::NQ) : "memory");
      const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char *>(p.y) + (size_t)clip * clip_rows * (C * 2), 0, (int)(clip_rows * (C * 2)), 0x00020000);
      unsigned einv[L::NQ];
#pragma unroll
      for (int q = 0; q < L::NQ; ++q) einv[q] = (unsigned)p0 + epx[q] < (unsigned)HW ? 0u : kInvalid;

      for (int nc0 = 0; nc0 < L::NC; nc0 += L::RD) {
#pragma unroll
        for (int rset = 0; rset < L::RD; ++rset) {
          const int nc = nc0 + rset;
          const bool last = nc + 1 == L::NC;
          C31P_STAMP(4);                                                   // (the tail of the previous slot)
          __builtin_amdgcn_s_barrier();                                    // A: W3's chunk nc is in LDS; LDS tile buffer `ybuf` is free
          C31P_STAMP(0);
          // ---- GEMM1: y[this wave's 32 rows][the chunk's 64 channels] ----
          f32x16 acc1[2];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
          {
            unsigned ba = w3a;
            asm volatile("" : "+v"(ba));                   // (keeps the KT1 addresses from being hoisted out of the chunk loop)
            constexpr int NM = L::KT1 * 2, D = 4;
            u32x4 ring[D];
#pragma unroll
            for (int m = 0; m < D; ++m)
              ring[m] = *reinterpret_cast<const u32x4 *>(lds + (ba ^ (unsigned)((m / 2) << 5)) + (m % 2) * 32 * L::RB3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              const u32x4 b = ring[m % D];
              if (m + D < NM)
                ring[m % D] = *reinterpret_cast<const u32x4 *>(lds + (ba ^ (unsigned)(((m + D) / 2) << 5)) + ((m + D) % 2) * 32 * L::RB3);
              acc1[m % 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[m / 2]), __builtin_bit_cast(bf16x8, b),
                                                                    acc1[m % 2], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          C31P_STAMP(1);
          __builtin_amdgcn_s_barrier();                                    // B: every producer has read W3's chunk
          C31P_STAMP(2);
          if (last) load_afr(nclip, np0, next_dead);                       // GEMM1 of this tile is over: the next tile's A fragments
          // ---- epilogue of the chunk: + bias3, + residual, ReLU, bf16 -> y (global) and the LDS tile ----
          const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(bias3_l + nc * 64 + c8 * 8);
          const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(bias3_l + nc * 64 + c8 * 8 + 4);
          const bool wrap = last || nc < L::RD - 1;                        // a tile's last chunk (its AF loads) lies in (nc - RD, nc]
          f32x4 cpair[2][2];
          // (two epilogue steps -- 16 tile rows -- per trip through the sub-slab: a producer is alone with its LDS latencies, the
          //  consumer on its SIMD is busy with its own stream; the in-kernel stamps put 8 round trips per chunk at a fifth of the slot)
#pragma unroll
          for (int q = 0; q < L::NQ; ++q) {
            static_assert(L::NQ % 2 == 0, "steps go through the sub-slab in pairs");
            if ((q & 1) == 0) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                  for (int r = 0; r < 4; ++r) Cw[(8 * qq + 4 * half + r) * 68 + j * 32 + l31] = acc1[j][4 * (q + qq) + r];
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (same wave wrote it: no barrier needed)
#pragma unroll
              for (int qq = 0; qq < 2; ++qq) {
                cpair[qq][0] = *reinterpret_cast<const f32x4 *>(Cw + (8 * qq + r8l) * 68 + c8 * 8);
                cpair[qq][1] = *reinterpret_cast<const f32x4 *>(Cw + (8 * qq + r8l) * 68 + c8 * 8 + 4);
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // read before the next pair overwrites it
            }
            const f32x4 c0 = cpair[q & 1][0], c1 = cpair[q & 1][1];
            // this step's residual (requested RD chunks ago)
            static_assert(L::wait_res(L::NC - 1) == L::wait_res(0) && L::wait_res(L::RD - 1) == L::wait_res(L::NC - 2), "the two cases below");
            C31P_STAMP(3);
            if (wrap) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rres[rset * L::NQ + q]) : "n"(L::wait_res(0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rres[rset * L::NQ + q]) : "n"(L::wait_res(L::RD - 1)) : "memory");
            C31P_STAMP(5);
            float v[8] = {c0[0] + bias0[0], c0[1] + bias0[1], c0[2] + bias0[2], c0[3] + bias0[3],
                          c1[0] + bias1[0], c1[1] + bias1[1], c1[2] + bias1[2], c1[3] + bias1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += split_elem(rres[rset * L::NQ + q], e);
            u32x4 o;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) o[w2] = pack_bf16(fmaxf(v[2 * w2], 0.f), fmaxf(v[2 * w2 + 1], 0.f));
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)(evoff[q] | einv[q]), p0 * cb + nc * 128, TSM_AUX_C31);
            *reinterpret_cast<u32x4 *>(lds + yw[q] + ybuf) = o;
            if (nc + L::RD < L::NC) load_res(rset * L::NQ + q, q, clip, p0, nc + L::RD, 0u);
            else load_res(rset * L::NQ + q, q, nclip, np0, nc + L::RD - L::NC, next_dead);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's rows of the LDS tile are written (before the next A)
          C31P_STAMP(3);
          ybuf ^= (unsigned)(L::M * 128);
        }
      }
      clip = nclip;
      p0 = np0;
    }
    __builtin_amdgcn_s_barrier();                                          // the drain slot: the consumers multiply the last chunk
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the dead loads of the tail)
#if TSM_C31P_STAMP
    if (bid == 0 && tid == 0)
      printf("c31p P K3=%d slots=%d: waitA %llu gemm1 %llu waitB %llu epilogue %llu (of it res waits %llu) tail %llu cycles\n", K3, nslots,
             stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3] + stamp_acc[5], stamp_acc[5], stamp_acc[4]);
#endif
  } else {
    // ================================ CONSUMERS: the weight stream, GEMM2, the t1 epilogue ================================
    // GEMM2's A fragments: tile row r = 32 rg + l31 as it stands, or rows r + PX / r - PX (frames t + 1 / t - 1), or zeros
    unsigned ybase[3], yflip[3];
    const int arow = 32 * rg + l31;
    {
      const int t = arow >> lpx, rp = arow + PX, rm = arow - PX;
      ybase[0] = (unsigned)(L::kY + arow * 128);   yflip[0] = (unsigned)((arow >> 1) & 7);
      ybase[1] = t + 1 < T ? (unsigned)(L::kY + rp * 128) : (unsigned)L::kZero;   yflip[1] = t + 1 < T ? (unsigned)((rp >> 1) & 7) : 0u;
      ybase[2] = t > 0 ? (unsigned)(L::kY + rm * 128) : (unsigned)L::kZero;       yflip[2] = t > 0 ? (unsigned)((rm >> 1) & 7) : 0u;
    }
    const unsigned w1flip = (unsigned)((l31 >> 1) & 7);
    const unsigned w1a = (unsigned)(L::kW1 + l31 * 128) + (((unsigned)half ^ (w1flip & 1u)) << 4) + ((w1flip >> 1) << 5);
    // DMA sources (the 16-byte chunk a lane fetches is swizzled on the SOURCE side; the LDS side is linear per piece)
    unsigned w3off[L::NW3], w1off[L::NW1];
#pragma unroll
    for (int i = 0; i < L::NW3; ++i) {
      const int row = (rg * L::NW3 + i) * L::RPP3 + lane / L::LPR3, slot = lane % L::LPR3;
      w3off[i] = (unsigned)(row * L::RB3 + ((slot ^ (row & 15)) << 4));
    }
#pragma unroll
    for (int i = 0; i < L::NW1; ++i) {
      const int row = (rg * L::NW1 + i) * 8 + (lane >> 3), slot = lane & 7;
      w1off[i] = (unsigned)(row * (C * 2) + ((slot ^ ((row >> 1) & 7)) << 4));
    }
    const __amdgpu_buffer_rsrc_t rsrcW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w3), 0, C * K3 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, N1 * C * 2, 0x00020000);
    auto w3_piece = [&](int nc, unsigned dead, int i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW3, (lds_void *)(lds + L::kW3 + (rg * L::NW3 + i) * 1024), 16,
                                               (int)(w3off[i] | dead), nc * 64 * rb3, 0, 0);
    };
    auto w1_piece = [&](int nc, unsigned buf, unsigned dead, int i) {    // buf: byte offset of the W1 buffer
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW1, (lds_void *)(lds + L::kW1 + buf + (rg * L::NW1 + i) * 1024), 16,
                                               (int)(w1off[i] | dead), nc * 128, 0, 0);
    };
    auto issue_w3 = [&](int nc, unsigned dead) {
#pragma unroll
      for (int i = 0; i < L::NW3; ++i) w3_piece(nc, dead, i);
    };
    // TSM_C31P_SPREAD: W1's pieces ride behind the GEMM2 MFMAs in front of barrier B, W3's behind the ones after it (a slot
    // without a chunk to multiply issues them at once); the default issues each set in one burst
    constexpr int NMq = 4 * L::NTL2;
    constexpr bool SPREAD = TSM_C31P_SPREAD != 0;
    constexpr int W1E = SPREAD && L::QSPLIT >= L::NW1 ? L::QSPLIT / L::NW1 : 0, W1D = W1E > 0 ? W1E : 1;               // MFMAs per W1 piece
    constexpr int W3E = SPREAD && NMq - L::QSPLIT >= L::NW3 ? (NMq - L::QSPLIT) / L::NW3 : 0, W3D = W3E > 0 ? W3E : 1;   // ... per W3 piece
    f32x16 acc2[L::NTL2];
#pragma unroll
    for (int j = 0; j < L::NTL2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;

    issue_w3(0, 0u);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                          // P0

    int kp = 0, sp = 0;                                                    // the producers' chunk / tile of this slot
    unsigned par = 0u;                                                     // slot & 1
    for (int g = 0; g <= nslots; ++g) {
      const bool live_p = g < nslots;
      const int kc = kp == 0 ? L::NC - 1 : kp - 1;                         // the chunk this slot multiplies (g >= 1)
      const bool have = g >= 1;
      C31P_STAMP(4);
      __builtin_amdgcn_s_barrier();                                        // A: LDS tile (g - 1) & 1 is written, W3 / W1 pieces of the last slot have landed
      C31P_STAMP(0);
      const unsigned w1buf = par * (unsigned)(N1 * 128), w1dead = live_p ? 0u : kInvalid;   // W1's chunk kp -> buffer g & 1 (free: read two slots ago)
      // ---- GEMM2 of chunk kc: t1 += shift(y chunk) * W1[:, chunk]^T, from LDS tile / W1 buffer (g - 1) & 1 ----
      const unsigned prev = (par ^ 1u);
      const int c0ch = kc * 64;
      const int sel = c0ch < p.fold ? 1 : (c0ch < 2 * p.fold ? 2 : 0);     // wave-uniform: frames t + 1 / t - 1 / t
      const unsigned yb0 = sel == 1 ? ybase[1] : sel == 2 ? ybase[2] : ybase[0];
      const unsigned yb = yb0 + ((yb0 >= (unsigned)L::kZero) ? 0u : prev * (unsigned)(L::M * 128));   // (the zero row has one copy)
      const unsigned yf = sel == 1 ? yflip[1] : sel == 2 ? yflip[2] : yflip[0];
      unsigned ya = yb + (((unsigned)half ^ (yf & 1u)) << 4) + ((yf >> 1) << 5), bb = w1a + prev * (unsigned)(N1 * 128);
      asm volatile("" : "+v"(ya), "+v"(bb));
      constexpr int NM = 4 * L::NTL2, D = 4;
      u32x4 a4[4], ring[D];
      if (have) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) a4[gg] = *reinterpret_cast<const u32x4 *>(lds + (ya ^ (unsigned)(gg << 5)));
#pragma unroll
        for (int m = 0; m < D; ++m)
          ring[m] = *reinterpret_cast<const u32x4 *>(lds + (bb ^ (unsigned)((m / L::NTL2) << 5)) + (m % L::NTL2) * 32 * 128);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < L::QSPLIT; ++m) {
          const u32x4 b = ring[m % D];
          if (m + D < NM)
            ring[m % D] = *reinterpret_cast<const u32x4 *>(lds + (bb ^ (unsigned)(((m + D) / L::NTL2) << 5)) + ((m + D) % L::NTL2) * 32 * 128);
          acc2[m % L::NTL2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4[m / L::NTL2]), __builtin_bit_cast(bf16x8, b),
                                                                      acc2[m % L::NTL2], 0, 0, 0);
          if (W1E > 0 && m % W1D == 0 && m / W1D < L::NW1) w1_piece(kp, w1buf, w1dead, m / W1D);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!have || W1E == 0) {
#pragma unroll
        for (int i = 0; i < L::NW1; ++i) w1_piece(kp, w1buf, w1dead, i);
      }
      C31P_STAMP(1);
      __builtin_amdgcn_s_barrier();                                        // B: the producers have read W3's chunk kp
      C31P_STAMP(2);
      const int kn = kp + 1 == L::NC ? 0 : kp + 1;                         // W3's chunk of the NEXT slot
      const unsigned w3dead = g + 1 < nslots ? 0u : kInvalid;
      if (!have || W3E == 0) issue_w3(kn, w3dead);
      if (have) {
#pragma unroll
        for (int m = L::QSPLIT; m < NM; ++m) {
          const u32x4 b = ring[m % D];
          if (m + D < NM)
            ring[m % D] = *reinterpret_cast<const u32x4 *>(lds + (bb ^ (unsigned)(((m + D) / L::NTL2) << 5)) + ((m + D) % L::NTL2) * 32 * 128);
          acc2[m % L::NTL2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4[m / L::NTL2]), __builtin_bit_cast(bf16x8, b),
                                                                      acc2[m % L::NTL2], 0, 0, 0);
          if (W3E > 0 && (m - L::QSPLIT) % W3D == 0 && (m - L::QSPLIT) / W3D < L::NW3) w3_piece(kn, w3dead, (m - L::QSPLIT) / W3D);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // the fragment reads of this slot are over (before the next A)
      C31P_STAMP(3);
      if (have && kc == L::NC - 1) {
        // ---- t1 of the tile the producers finished a slot ago: relu(acc2 + bias1) -> bf16, whole 128-byte row segments ----
        int tclip, tp0;
        tile_of(kp == 0 ? sp - 1 : sp, &tclip, &tp0);
        const __amdgpu_buffer_rsrc_t rsrcT1 = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char *>(p.t1) + (size_t)tclip * clip_rows * (N1 * 2), 0, (int)(clip_rows * (N1 * 2)), 0x00020000);
#pragma unroll
        for (int jh = 0; jh < L::NTL2 / 2; ++jh) {
          const int col0 = jh * 64;
          const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias1_l + col0 + c8 * 8);
          const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias1_l + col0 + c8 * 8 + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) Cw[(4 * half + r) * 68 + j * 32 + l31] = acc2[2 * jh + j][4 * q + r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const f32x4 c0 = *reinterpret_cast<const f32x4 *>(Cw + r8l * 68 + c8 * 8);
            const f32x4 c1 = *reinterpret_cast<const f32x4 *>(Cw + r8l * 68 + c8 * 8 + 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float v[8] = {c0[0] + b0[0], c0[1] + b0[1], c0[2] + b0[2], c0[3] + b0[3],
                                c1[0] + b1[0], c1[1] + b1[1], c1[2] + b1[2], c1[3] + b1[3]};
            u32x4 o;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) o[w2] = pack_bf16(fmaxf(v[2 * w2], 0.f), fmaxf(v[2 * w2 + 1], 0.f));
            const int row = 32 * rg + 8 * q + r8l, t = row >> lpx, px = row & (PX - 1);
            const unsigned off = (unsigned)((t * HW + px) * (N1 * 2) + c8 * 16), inv = tp0 + px < HW ? 0u : kInvalid;
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcT1, (int)(off | inv), tp0 * n1b + col0 * 2, TSM_AUX_C31);
          }
        }
#pragma unroll
        for (int j = 0; j < L::NTL2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
        wait_vmcnt_imm<L::NT1S>();                                         // this slot's DMA pieces have landed (the t1 stores are younger)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this slot's DMA pieces have landed
      }
      C31P_STAMP(5);
      par ^= 1u;
      if (++kp == L::NC) { kp = 0; ++sp; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if TSM_C31P_STAMP
    if (bid == 0 && tid == 256)
      printf("c31p Q K3=%d slots=%d: waitA %llu w1+mfma0 %llu waitB %llu w3+mfma1 %llu t1+dma wait %llu tail %llu cycles\n", K3, nslots,
             stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3], stamp_acc[5], stamp_acc[4]);
#endif
  }
}

// Instantiations: (K3, C, N1) = (128, 512, 128) layer2.k -> layer2.k+1 with CH = 1 (tiles of 256 rows); (128, 512, 256)
// layer2.3 -> layer3.0 and (256, 1024, 256) layer3.k -> layer3.k+1 with CH = 2 (wave pairs, tiles of 128 rows).
static int conv31_rows(const Conv31Params &p) {
  if (p.K3 == 128 && p.C == 512 && p.N1 == 128) return 256;
  if ((p.K3 == 128 && p.C == 512 && p.N1 == 256) || (p.K3 == 256 && p.C == 1024 && p.N1 == 256)) return 128;
  return 0;
}

bool conv31_valid(const Conv31Params &p) {
  const int m = conv31_rows(p);
  if (m == 0) return false;
  if (p.n_clips <= 0 || p.HW <= 0 || p.T <= 0 || m % p.T != 0 || m / p.T < 8) return false;
  if (p.fold != 0 && (p.fold % 64 != 0 || 2 * p.fold > p.C)) return false;  // a 64-channel chunk is shifted as a whole
  return (double)p.T * p.HW * p.C * 2.0 < 2.0e9;                             // 32-bit offsets inside a clip's block
}

template <int K3, int C, int N1, int CH>
static hipError_t launch_c31(const Conv31Params &p, long ntiles, int n_cu, hipStream_t s) {
  constexpr size_t kLdsBytes = C31<K3, C, N1, CH>::kBytes;
  const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(512);
  TSM_KLAUNCH((conv31_fused_kernel<K3, C, N1, CH>), grid, block, kLdsBytes, s, p);
  return hipGetLastError();
}

#ifndef TSM_C31_PC
#define TSM_C31_PC 1
#endif
template <int K3, int C, int N1>
static hipError_t launch_c31p(const Conv31Params &p, long ntiles, int n_cu, hipStream_t s) {
  constexpr size_t kLdsBytes = C31P<K3, C, N1>::kBytes;
  const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(512);
  TSM_KLAUNCH((conv31_pc_kernel<K3, C, N1>), grid, block, kLdsBytes, s, p);
  return hipGetLastError();
}

hipError_t launch_conv31_fused(const Conv31Params &p_in, hipStream_t s) {
  Conv31Params p = p_in;
  if (!p.t2 || !p.w3 || !p.bias3 || !p.res || !p.y || !p.w1 || !p.bias1 || !p.t1 || !conv31_valid(p)) return hipErrorInvalidValue;
  const int px = conv31_rows(p) / p.T;
  p.log_px = 0;
  while ((1 << p.log_px) < px) ++p.log_px;
  const DeviceInfo &di = device_info();
  if (di.status != hipSuccess) return di.status;
  const long ntiles = (long)p.n_clips * ((p.HW + px - 1) / px);
  if (p.K3 == 128 && p.N1 == 128) return launch_c31<128, 512, 128, 1>(p, ntiles, di.n_cu, s);
#if TSM_C31_PC     // the 128-row tiles on the producer / consumer form (0: round 4's wave pairs, for A/B builds)
  if (p.K3 == 128) return launch_c31p<128, 512, 256>(p, ntiles, di.n_cu, s);
  return launch_c31p<256, 1024, 256>(p, ntiles, di.n_cu, s);
#else
  if (p.K3 == 128) return launch_c31<128, 512, 256, 2>(p, ntiles, di.n_cu, s);
  return launch_c31<256, 1024, 256, 2>(p, ntiles, di.n_cu, s);
#endif
}

hipError_t opt_in_conv31() {
  hipError_t first = hipSuccess;
  auto opt_in = [&](const void *fn, size_t bytes) {
    const hipError_t st = lds_opt_in(fn, bytes);
    if (st != hipSuccess && first == hipSuccess) first = st;
  };
  opt_in(reinterpret_cast<const void *>(&conv31_fused_kernel<128, 512, 128, 1>), C31<128, 512, 128, 1>::kBytes);
  opt_in(reinterpret_cast<const void *>(&conv31_fused_kernel<128, 512, 256, 2>), C31<128, 512, 256, 2>::kBytes);
  opt_in(reinterpret_cast<const void *>(&conv31_fused_kernel<256, 1024, 256, 2>), C31<256, 1024, 256, 2>::kBytes);
  opt_in(reinterpret_cast<const void *>(&conv31_pc_kernel<128, 512, 256>), C31P<128, 512, 256>::kBytes);
  opt_in(reinterpret_cast<const void *>(&conv31_pc_kernel<256, 1024, 256>), C31P<256, 1024, 256>::kBytes);
  return first;
}

}  // namespace tsm
