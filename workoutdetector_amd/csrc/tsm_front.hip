// front_s2_kernel: temporal shift + conv1 (1x1, 256 -> 128) + conv2 (3x3, stride 2, 128 -> 128) of layer2.0 in ONE launch, bf16.
#include "tsm_device.h"

namespace tsm {

// ---------------------------------------------------------------------------------------------
// Why (round 5): the first block of layer2 is the most expensive block of the config-5 forward after layer1's -- conv1 streams
// the 2.15-GB block input and WRITES a 1.07-GB mid tensor t1 (647 us at 5.0 TB/s: on the HBM roof), the stride-2 3x3 reads it
// back (316 us) -- and neither launch has anything left to give on its own.  The only lever is the bytes between them: here t1
// never exists in memory.  Per frame: read H * W * 512 B, write Ho * Wo * 256 B (the two launches: 2.4 x that).
//
// Structure (bneck_ws_kernel's line buffer, conv3x3_ws128_kernel<true>'s stride-2 reads): one persistent workgroup of four waves
// per CU (one wave per SIMD, the whole register file), WHOLE FRAMES per workgroup, walked top to bottom one input ROW at a time:
//   * the row's W <= 64 pixels x 256 channels (32 KB) arrive by LDS-DMA in one of THREE row slots, requested three rows ahead:
//     64 KB of the block input are in flight per CU while a row is multiplied; wave w fills k16 planes 4 w .. 4 w + 3 (the
//     temporal shift is the choice of source frame for planes 0-3: channels 0-31 from t + 1, 32-63 from t - 1, zeros at the
//     clip's ends); plane = 32 bytes per pixel, the two 16-byte halves swapped where (pixel >> 3) is odd (source side);
//   * conv1: the OUTPUT CHANNELS are split over the waves (wave w keeps W1's rows 32 w .. + 31: 16 fragments) and every wave
//     multiplies all pixels of the row (two M-tiles of 32), transposed (A = weights, B = pixels): a lane ends up with one pixel
//     and 4-channel groups -> bias1, ReLU, bf16, v_permlane32_swap -> 16-byte groups = halves of its entry in planes 2 w, 2 w + 1
//     of a three-row LINE BUFFER in LDS (t1: 128 channels = 8 planes).  A row of the line buffer is stored with its columns
//     DE-INTERLEAVED -- padded column p = c + 1: the even p first (E run), then the odd p (O run) -- so that the stride-2 taps
//     kx = 0 / 1 / 2 of output column ox read positions E[ox] / O[ox] / E[ox + 1]: consecutive lanes, consecutive positions,
//     conflict-free with the bit-3 swap; p = 0 and p = W + 1 (conv2's padding) are never written and stay zero;
//   * conv2, after every odd row r = 2 s + 1: output row s (Wo <= 32 pixels = ONE M-tile) from line-buffer rows 2 s - 1, 2 s, 2 s + 1
//     (row -1 = a zeroed slot), wave w's 32 output channels against its W2 slice in registers (72 fragments, 56 of them in
//     accumulation registers), one pixel-fragment read per MFMA; bias2, ReLU, bf16, two 16-byte stores per lane.
// Two barriers per row; every vector-memory wait is a counted vmcnt over the fixed issue order of a wave
//     even row r: nothing;  odd row r + 1: [the 8 LDS-DMA pieces of row r + 3: four between conv1's MFMAs, four between conv2's first]
//     [the 8 pieces of row r + 4, between conv2's MFMAs] [2 stores]
// (no register loads inside the loop: hipcc inserts no waits of its own).
// Products enter every accumulator in the separate kernels' order (conv1: k16 groups ascending; conv2: taps, then k16 groups;
// a * b commutes) and the two epilogues are conv1x1_wsn's and conv3x3_ws128's: bit-identical to the two launches it replaces.
// Needs W <= 64, H even (no zero row below the frame), fold = 32.
// ---------------------------------------------------------------------------------------------
constexpr int kFrRP = 68;                          // line-buffer positions per row: E run at 0 (<= 34), O run at 34
constexpr int kFrLbPlane = 3 * kFrRP * 32;         // one k16 group of the three buffered rows: 6 528 B
constexpr int kFrLbBytes = 8 * kFrLbPlane;         // 52 224 B
constexpr int kFrXOff = kFrLbBytes;                // three input row slots of 16 planes x 64 pixels x 32 B
constexpr int kFrSlot = 16 * 2048;
constexpr int kFrBiasOff = kFrXOff + 3 * kFrSlot;
constexpr int kFrBytes = kFrBiasOff + 2 * 128 * 4; // 151 552 B
constexpr int kFrAgprFrags = 56;
#ifndef TSM_FRONT_C1_PIECES
#define TSM_FRONT_C1_PIECES 8
#endif
constexpr int kFrC1Pieces = TSM_FRONT_C1_PIECES;   // pieces of the even row's refill that ride on conv1's 32 MFMAs (the rest on conv2's 72)

#ifndef TSM_FRONT_STAMP
#define TSM_FRONT_STAMP 0   // diagnostic builds only: per-phase cycle sums of workgroup 0, wave 0 (s_memtime), printed at the kernel's end
#endif
#if TSM_FRONT_STAMP
#define FRONT_STAMP(i)                                      \
  do {                                                      \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += _t - stamp_last;                        \
    stamp_last = _t;                                        \
  } while (0)
#else
#define FRONT_STAMP(i) do {} while (0)
#endif
template <bool SHIFT>
__global__ void __launch_bounds__(256, 1) front_s2_kernel(const FrontParams p) {
#if TSM_FRONT_STAMP
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W, Ho = H / 2, Wo = (W - 1) / 2 + 1;
  const int xframe = H * W * 512, yframe = Ho * Wo * 256;

  // ---- the stationary operands: this wave's 32 mid channels of W1, its 32 output channels of W2
  const __amdgpu_buffer_rsrc_t rsrcW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, 128 * 256 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, 128 * 1152 * 2, 0x00020000);
  u32x4 w1r[16], w2r[72];
#pragma unroll
  for (int s = 0; s < 72; ++s)
    w2r[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW2, ((wave * 32 + l31) * 1152 + s * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int g = 0; g < 16; ++g)
    w1r[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW1, ((wave * 32 + l31) * 256 + g * 16 + half * 8) * 2, 0, 0);
#pragma unroll
  for (int s = 0; s < kFrAgprFrags; ++s) asm volatile("" : "+a"(w2r[s]));
  float *bias1_lds = reinterpret_cast<float *>(lds + kFrBiasOff), *bias2_lds = bias1_lds + 128;
  if (tid < 128) {
    bias1_lds[tid] = p.bias1[tid];
    bias2_lds[tid] = p.bias2[tid];
  }
  for (int i = tid; i < kFrLbBytes / 16; i += 256) *reinterpret_cast<u32x4 *>(lds + i * 16) = u32x4{0u, 0u, 0u, 0u};

  // ---- lane constants
  // loader: piece (pl, j) of a row = pixels 32 j .. + 31 of plane 4 wave + pl; this lane fills half (lane & 1) of pixel 32 j + (lane >> 1)
  const int hsel = (lane & 1) ^ ((lane >> 4) & 1);
  // conv1: pixel 32 mt + l31 of the row; its fragment in a slot plane, and its two 16-byte writes into the line buffer
  unsigned xrd[2], lbw[2];
  bool ok1[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int c = 32 * mt + l31;
    xrd[mt] = (unsigned)(c * 32 + ((half ^ ((c >> 3) & 1)) << 4));
    const int pp = c + 1, pos = (pp & 1) ? 34 + (pp >> 1) : (pp >> 1);
    lbw[mt] = (unsigned)(pos * 32 + ((half ^ ((pos >> 3) & 1)) << 4));
    ok1[mt] = c < W;
  }
  // conv2: output column l31; positions of its three taps kx in a line-buffer row (E[ox], O[ox], E[ox + 1])
  unsigned lbr[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int pos = l31 + (kx == 1 ? 34 : (kx >> 1));
    lbr[kx] = (unsigned)(pos * 32 + ((half ^ ((pos >> 3) & 1)) << 4));
  }
  const bool ok2 = l31 < Wo;
  const float floor_ = 0.f;

  // virtual frame index -> frame: XCD-chunked (the shifted channels come from frames f - 1 and f + 1), then the engine's direction
  auto frame_of = [&](int v) {
    const int n = p.N, q8 = n >> 3, r8 = n & 7, x = v & 7;
    const int c = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (v >> 3);
    return p.reverse ? p.N - 1 - c : c;
  };
  // LDS-DMA of a row into a row slot, as eight pieces that can be issued one at a time (between MFMAs: a piece costs ~100 cycles of
  // issue in a burst behind a barrier -- a sixth of the kernel, in-kernel stamps -- and about nothing beside a matrix instruction):
  // piece k = pixels 32 (k >> 2) .. + 31 of plane 4 wave + (k & 3).  Always eight operations per row (dead ones fetch nothing).
  struct RowDma {
    const char *base;      // one frame BEFORE the row's frame (only ever addressed there when frame t - 1 exists)
    unsigned v01[2], v23[2];
    unsigned char *dst;
  };
  auto prep_row = [&](int f, int r, int slot, bool live) -> RowDma {
    RowDma d;
    const int tt = p.T > 0 ? f % p.T : 0;
    d.base = reinterpret_cast<const char *>(p.x) + ((long)f - 1) * xframe;
    d.dst = lds + kFrXOff + slot * kFrSlot + 4 * wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 32 * j + (lane >> 1);
      const bool okp = live && c < W;
      const unsigned own = (unsigned)xframe + (unsigned)((r * W + c) * 512 + 4 * wave * 32 + hsel * 16);
      const unsigned vC = okp ? own : kInvalid;
      const unsigned vA = (okp && tt < p.T - 1) ? own + (unsigned)xframe : kInvalid;
      const unsigned vB = (okp && tt > 0) ? own - (unsigned)xframe : kInvalid;
      // fold = 32 channels: planes 0, 1 from frame t + 1, planes 2, 3 from t - 1 (all of them wave 0's)
      d.v01[j] = (SHIFT && wave == 0) ? vA : vC;
      d.v23[j] = (SHIFT && wave == 0) ? vB : vC;
    }
    return d;
  };
  auto piece = [&](const RowDma &d, int k) {
    const int j = k >> 2, pl = k & 3;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(d.base), 0, 3 * xframe, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(d.dst + pl * 2048 + j * 1024), 16, (int)(pl < 2 ? d.v01[j] : d.v23[j]), pl * 32, 0, 0);
  };

  // the (frame, row) the loader is at: up to three rows ahead of the row being multiplied
  int fi = blockIdx.x;                   // virtual index of the frame being multiplied
  int lfi = fi, lrow = 0;                // ... of the frame / row being requested
  auto prep_next = [&](int slot) -> RowDma {
    const RowDma d = prep_row(lfi < p.N ? frame_of(lfi) : 0, lrow, slot, lfi < p.N);
    if (++lrow == H) {
      lrow = 0;
      lfi += (int)gridDim.x;
    }
    return d;
  };
  auto issue_next = [&](int slot) {
    const RowDma d = prep_next(slot);
#pragma unroll
    for (int k = 0; k < 8; ++k) piece(d, k);
  };
  issue_next(0);
  issue_next(1);
  issue_next(2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // weights, biases, the zeroed line buffer, the first three rows

  int xslot = 0;                         // input ring slot of the row being multiplied (runs on across frames)
  for (; fi < p.N; fi += gridDim.x) {
    const int f = frame_of(fi);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.y) + (size_t)f * yframe, 0, yframe, 0x00020000);
    int lslot = 1;                       // line-buffer slot of row r: (r + 1) % 3 (row -1 lives in slot 0)
    // conv1 of the row in input slot `xs_slot` -> line-buffer slot `ls`; the eight pieces of `d` (if any) ride between its MFMAs.
    // Four batches of eight k16 groups -- (M-tile 0, groups 0-7), (0, 8-15), (1, 0-7), (1, 8-15) -- software-pipelined: the fragment
    // reads of batch i + 1 are issued in front of the MFMAs of batch i, and M-tile 0's epilogue rides on M-tile 1's first batch
    // (a wave is alone on its SIMD: nothing else hides an LDS round trip or the epilogue's ALU work).
    auto conv1_row = [&](int xs_slot, int ls, bool with_dma, const RowDma &d) {
      const unsigned char *xs = lds + kFrXOff + xs_slot * kFrSlot;
      u32x4 xf[2][8];
      auto load_batch = [&](int i) {
#pragma unroll
        for (int g = 0; g < 8; ++g) xf[i & 1][g] = *reinterpret_cast<const u32x4 *>(xs + xrd[i >> 1] + (8 * (i & 1) + g) * 2048);
      };
      // bias1, ReLU, bf16; the swap pairs groups (0, 1) and (2, 3): this lane then holds channels 16 g' + 8 half .. + 8 of
      // k16 group g' = 2 wave + qq of its pixel = one 16-byte half of the pixel's entry in plane g' of the line buffer
      auto epilogue = [&](const f32x16 &acc, int mt) {
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(bias1_lds + wave * 32 + 8 * q + 4 * half);
          pk[q][0] = pack_bf16(fmaxf(acc[4 * q] + b[0], floor_), fmaxf(acc[4 * q + 1] + b[1], floor_));
          pk[q][1] = pack_bf16(fmaxf(acc[4 * q + 2] + b[2], floor_), fmaxf(acc[4 * q + 3] + b[3], floor_));
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][0], pk[2 * qq + 1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * qq][1], pk[2 * qq + 1][1], false, false);
          if (ok1[mt])
            *reinterpret_cast<u32x4 *>(lds + (2 * wave + qq) * kFrLbPlane + ls * (kFrRP * 32) + lbw[mt]) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      };
      f32x16 acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc0[e] = 0.f;
        acc1[e] = 0.f;
      }
      load_batch(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) load_batch(i + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (i < 2) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[8 * (i & 1) + g]), __builtin_bit_cast(bf16x8, xf[i & 1][g]), acc0, 0, 0, 0);
          else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[8 * (i & 1) + g]), __builtin_bit_cast(bf16x8, xf[i & 1][g]), acc1, 0, 0, 0);
          // kFrC1Pieces / 4 pieces per batch, BEHIND an MFMA (in front of them the wave would stall on its issue first)
          if (with_dma && (g == 1 || (kFrC1Pieces == 8 && g == 5))) {
            __builtin_amdgcn_sched_barrier(0);
            piece(d, kFrC1Pieces == 8 ? 2 * i + (g >> 2) : i);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (i == 2) {      // M-tile 0's epilogue, spread over these eight MFMAs
          epilogue(acc0, 0);
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // VALU
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      epilogue(acc1, 1);
    };
    for (int r = 0; r < H; r += 2) {
      // ================= even row r: conv1 -> line-buffer slot lslot (its input slot is re-armed from inside the next row) =================
      // this wave's pieces of row r have landed (they were the SECOND group of row r - 3): younger are that row's two stores and
      // all of row r - 1's operations (8 + 8 + 2)
      FRONT_STAMP(5);
      asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      FRONT_STAMP(0);
      __builtin_amdgcn_s_barrier();      // ... every wave's; nobody still reads the line buffer (conv2 of the previous row pair is over)
      FRONT_STAMP(1);
      if (r == 0) {                      // row -1 of the new frame: zeros in slot 0 (the previous frame's row H - 2 lived there)
        for (int i = tid; i < 8 * kFrRP * 2; i += 256) {
          const int pl = i / (kFrRP * 2), q = i - pl * (kFrRP * 2);
          *reinterpret_cast<u32x4 *>(lds + pl * kFrLbPlane + q * 16) = u32x4{0u, 0u, 0u, 0u};
        }
      }
      const int xs_even = xslot, ls_even = lslot;
      {
        RowDma none{};
        conv1_row(xs_even, ls_even, false, none);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FRONT_STAMP(2);
      __builtin_amdgcn_s_barrier();      // the line-buffer row is complete; every wave has read input slot xs_even
      FRONT_STAMP(1);
      xslot = xslot == 2 ? 0 : xslot + 1;
      lslot = lslot == 2 ? 0 : lslot + 1;
      // ================= odd row r + 1: conv1 (+ the refill of slot xs_even), conv2 (+ the refill of its own slot) =================
      // its pieces were the FIRST group of row r - 1: younger are that row's second group and its two stores
      FRONT_STAMP(5);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      FRONT_STAMP(0);
      __builtin_amdgcn_s_barrier();
      FRONT_STAMP(1);
      const RowDma dA = prep_next(xs_even);
      conv1_row(xslot, lslot, true, dA);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      FRONT_STAMP(2);
      __builtin_amdgcn_s_barrier();      // rows r - 1, r, r + 1 of the line buffer are complete; every wave has read input slot xslot
      FRONT_STAMP(1);
      {
        const RowDma dB = prep_next(xslot);
        FRONT_STAMP(3);
        // ---- conv2: output row s = r / 2 from line-buffer rows r - 1, r, r + 1 ----
        const int s = r >> 1;
        // (slot of row r - 1 + ky: lslot is row r + 1's, i.e. ky = 2; ky = 1 -> lslot - 1, ky = 0 -> lslot - 2, mod 3)
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        u32x4 px[8];                     // pixel fragments, five steps ahead of their MFMA
        unsigned rb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          int sl = lslot + ky - 2;
          sl += sl < 0 ? 3 : 0;
          rb[ky] = (unsigned)(sl * (kFrRP * 32));
        }
        auto rd = [&](int st) {
          const int tap = st >> 3, g = st & 7, ky = tap / 3, kx = tap - ky * 3;
          px[st & 7] = *reinterpret_cast<const u32x4 *>(lds + rb[ky] + lbr[kx] + g * kFrLbPlane);
        };
        rd(0); rd(1); rd(2); rd(3); rd(4);
        static_for<72>([&](auto sc) __attribute__((always_inline)) {
          constexpr int st = decltype(sc)::value;
          if constexpr (st + 5 < 72) rd(st + 5);
          // the rest of the even row's refill (its first four pieces rode on conv1), then this row's: a piece every five or six MFMAs
          if constexpr (kFrC1Pieces == 4) {
            if constexpr (st < 24 && st % 6 == 3) piece(dA, 4 + st / 6);
            if constexpr (st >= 27 && st < 67 && (st - 27) % 5 == 0) piece(dB, (st - 27) / 5);
          } else {
            if constexpr (st % 9 == 4) piece(dB, st / 9);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2r[st]), __builtin_bit_cast(bf16x8, px[st & 7]), acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
        FRONT_STAMP(4);
        // bias2, ReLU, bf16; lanes 0-31 take groups 0, 1 and lanes 32-63 groups 2, 3 of the pixel: two 16-byte stores into this
        // wave's 64-byte slice of the pixel
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4 *>(bias2_lds + wave * 32 + 8 * q + 4 * half);
          pk[q][0] = pack_bf16(fmaxf(acc[4 * q] + b[0], floor_), fmaxf(acc[4 * q + 1] + b[1], floor_));
          pk[q][1] = pack_bf16(fmaxf(acc[4 * q + 2] + b[2], floor_), fmaxf(acc[4 * q + 3] + b[3], floor_));
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
          for (int w2 = 0; w2 < 2; ++w2) {
            const auto r2 = __builtin_amdgcn_permlane32_swap(pk[qq][w2], pk[qq + 2][w2], false, false);
            pk[qq][w2] = r2[0];
            pk[qq + 2][w2] = r2[1];
          }
        const unsigned yo = ok2 ? (unsigned)((s * Wo + l31) * 256 + wave * 64 + 2 * half * 16) : kInvalid;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const u32x4 o = {pk[qq][0], pk[qq][1], pk[qq + 2][0], pk[qq + 2][1]};
          __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)yo, qq * 16, TSM_AUX_WS);
        }
      }
      xslot = xslot == 2 ? 0 : xslot + 1;
      lslot = lslot == 2 ? 0 : lslot + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dead rows of the tail land before the workgroup leaves its LDS
#if TSM_FRONT_STAMP
  if (blockIdx.x == 0 && tid == 0)
    printf("front wave 0: DMA-landed wait %llu | barriers %llu | conv1 (MFMA + epilogue + its DMA pieces) %llu | DMA prep %llu | conv2 MFMA + its DMA pieces %llu | conv2 epilogue + stores + loop %llu cycles\n",
           stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3], stamp_acc[4], stamp_acc[5]);
#endif
}

bool front_s2_valid(int n, int h, int w, int T, int fold) {
  return n > 0 && h >= 2 && (h & 1) == 0 && w >= 2 && w <= 64 && (double)h * w * 512.0 * 3.0 < 2.0e9 &&
         (T == 0 || (T > 0 && n % T == 0 && fold == 32));
}

hipError_t launch_front_s2(const FrontParams &p, hipStream_t s) {
  if (!p.x || !p.w1 || !p.bias1 || !p.w2 || !p.bias2 || !p.y) return hipErrorInvalidValue;
  if (!front_s2_valid(p.N, p.H, p.W, p.T, p.fold)) return hipErrorInvalidValue;
  const DeviceInfo &di = device_info();
  if (di.status != hipSuccess) return di.status;
  const dim3 grid((unsigned)(p.N < di.n_cu ? p.N : di.n_cu)), block(256);
  if (p.T > 0) TSM_KLAUNCH(front_s2_kernel<true>, grid, block, kFrBytes, s, p);
  else TSM_KLAUNCH(front_s2_kernel<false>, grid, block, kFrBytes, s, p);
  return hipGetLastError();
}

hipError_t opt_in_front() {
  hipError_t first = lds_opt_in(reinterpret_cast<const void *>(&front_s2_kernel<true>), kFrBytes);
  const hipError_t st = lds_opt_in(reinterpret_cast<const void *>(&front_s2_kernel<false>), kFrBytes);
  return first != hipSuccess ? first : st;
}

}  // namespace tsm
