// bneck_ws_kernel: a whole layer1 Bottleneck (shift, conv1, conv2, conv3 + identity) in one launch, bf16.
#include "tsm_device.h"

namespace tsm {

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
  static constexpr int kW1pOff = kBytes;                        // IDL: k16 groups 8-15 of W1's second tile, fragment order (8 KB)
  static constexpr int kBytesIdl = kW1pOff + 8 * 1024;          // 158 208 B
};

// SHIFT: the temporal shift of conv1's input, fold = CIN / 8 channels from frame t + 1 and as many from t - 1 (the bf16
// formats take shift_div 8 only) -- compile-time, so that a chunk's source row is a register choice and not a table lookup.
// IDL (CIN = 256, W = 64): half of conv3's identity operand is taken from the input slots in LDS instead of a second read
// of the block input.  With full-width rows of 64 pixels a step's 128 input pixels are exactly the four waves' slots, and
// output row 2s (M-tiles 2, 3) is the FIRST row of this step's input = slots 0, 1: once every slot has landed (one more
// barrier at the top of a step) each wave copies its 64 output channels of those two slots into the registers the late
// identity loads used to fill, and the slots are re-armed behind the conv1 barrier instead of inside the phase.  Output row
// 2s - 1 (M-tiles 0, 1) was the second row of the PREVIOUS step's input: holding it for a step takes 32 registers the file
// does not have (built: 512 registers and 20 spills), so it still comes from memory.  With the temporal shift the first 64
// channels of a slot come from frames t +- 1, so the wave that owns output channels 0-63 keeps loading all of its identity.
#ifndef TSM_BNECK_PXR64
#define TSM_BNECK_PXR64 4   // conv2's fragment ring of the 64-channel form (8 = reads seven steps ahead instead of three: measured, no change)
#endif
#ifndef TSM_BNECK_X
#define TSM_BNECK_X 0       // timing probes (garbage results): 1 the output stores fully coalesced (1 KB contiguous per instruction),
#endif                      // 2 the identity loads of the 64-channel form likewise
#ifndef TSM_BNECK_STAMP
#define TSM_BNECK_STAMP 0   // diagnostic builds only: per-phase cycle sums of workgroup 0's four waves (s_memtime), printed at the kernel's end
#endif
#if TSM_BNECK_STAMP
#define BN_STAMP(i)                                         \
  do {                                                      \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += _t - stamp_last;                        \
    stamp_last = _t;                                        \
  } while (0)
#else
#define BN_STAMP(i) do {} while (0)
#endif
template <int CIN, bool SHIFT, bool IDL = false>
__global__ void __launch_bounds__(256, 1) bneck_ws_kernel(const BneckParams p) {
  constexpr bool DUAL = CIN == 64;
  static_assert(!IDL || CIN == 256, "the LDS identity is the 256-channel form's");
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
  const int H = p.H, W = IDL ? 64 : p.W, W2 = 2 * W;   // (IDL is launched for W = 64 only: a constant there)
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
  // IDL: the second tile's k16 groups 8-15 live in LDS (one copy for the four waves, read once per step): 32 registers
  // for the identity set that is held across a step
  constexpr int kW1pOff = BnLds<CIN>::kW1pOff;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < NG1; ++g) {
      if (IDL && nt == 1 && g >= 8) {
        if (wave == ((g - 8) >> 1))
          *reinterpret_cast<u32x4 *>(lds + kW1pOff + (g - 8) * 1024 + lane * 16) =
              __builtin_amdgcn_raw_buffer_load_b128(rsrcW1, ((nt * 32 + l31) * CIN + g * 16 + half * 8) * 2, 0, 0);
      } else {
        w1r[nt][g] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW1, ((nt * 32 + l31) * CIN + g * 16 + half * 8) * 2, 0, 0);
      }
    }
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

  // IDL: does THIS wave take its identity from LDS (wave-uniform), and where: 16-byte group 2 half + qq of the 64-byte slice
  // it = 2 wave + itl of pixel l31 of a slot = plane 2 it + half, half qq of the pixel's entry
  const bool idl = IDL && (!SHIFT || wave != 0);
  const unsigned idrd = (unsigned)(kBnXOff + (2 * (2 * wave) + half) * 1024 + l31 * 32);
  const unsigned idflip = (unsigned)((l31 >> 3) & 1);

  u32x4 nxt[IDL ? 2 : 1][4];
#pragma unroll
  for (int mt = 0; mt < (IDL ? 2 : 1); ++mt)
#pragma unroll
    for (int k = 0; k < 4; ++k) nxt[mt][k] = u32x4{0u, 0u, 0u, 0u};

  // LDS-DMA of the input of step s of frame f into this wave's slot: always NDMA operations (dead ones fetch nothing), which can be
  // issued one at a time (IDL form: behind conv2's MFMAs -- in a burst behind the conv1 barrier a piece costs ~100 cycles of issue)
  struct XDma {
    const char *base;      // one frame BEFORE f (only ever addressed there when frame t - 1 exists)
    unsigned vC, vA, vB;   // the pixel's row in its own frame, in frame t + 1, in frame t - 1
  };
  auto prep_x = [&](int f, int s, bool live) -> XDma {
    XDma d;
    const int tt = p.T > 0 ? f % p.T : 0;
    d.base = reinterpret_cast<const char *>(p.x) + ((long)f - 1) * xframe;
    const int pix = 2 * s * W + md;
    const bool okp = live && md < W2 && pix < H * W;
    // three source rows per pixel: its own frame, frame t + 1, frame t - 1; a k16 group's 32 bytes are the instruction's
    // immediate offset, so a step costs three address registers, not one per group
    const unsigned own = (unsigned)xframe + (unsigned)pix * (unsigned)XROW + (unsigned)hsel * 16u;
    d.vC = okp ? own : kInvalid;
    d.vA = (okp && tt < p.T - 1) ? own + (unsigned)xframe : kInvalid;
    d.vB = (okp && tt > 0) ? own - (unsigned)xframe : kInvalid;
    return d;
  };
  auto x_piece = [&](const XDma &d, auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(d.base), 0, 3 * xframe, 0x00020000);
    // fold = CIN / 8 channels: CIN 256 -- k16 groups 0, 1 from t + 1 and 2, 3 from t - 1; CIN 64 -- ONE 16-byte chunk each:
    // the two halves of k16 group 0 (the loader lane's hsel picks the chunk)
    const unsigned v0 = !SHIFT ? d.vC : (DUAL ? (hsel ? d.vB : d.vA) : d.vA);
    const unsigned v = !SHIFT ? d.vC : (DUAL ? (g == 0 ? v0 : d.vC) : (g < 2 ? d.vA : (g < 4 ? d.vB : d.vC)));
    // (the instruction's immediate offset is added to the LDS address as well as to the memory address: the LDS base
    //  carries g * 1024 - g * 32 so that plane g still starts at g * 1024)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(xs + g * (1024 - 32)), 16, (int)v, 0, g * 32, 0);
  };
  auto issue_x = [&](int f, int s, bool live) {
    const XDma d = prep_x(f, s, live);
    static_for<NG1>([&](auto gc) __attribute__((always_inline)) { x_piece(d, gc); });
  };

  // virtual frame index -> frame: XCD-chunked (the shifted quarter of conv1's input comes from frames f - 1 and f + 1: with
  // a contiguous eighth of the frames per XCD its neighbours' workgroups fetch the same lines at about the same time), then
  // the engine's alternating direction
  auto frame_of = [&](int v) {
    const int n = p.N, q8 = n >> 3, r8 = n & 7, x = v & 7;          // xcd_chunked() in 32 bits (scalar registers are short here)
    const int c = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (v >> 3);
    return p.reverse ? p.N - 1 - c : c;
  };
  int fi = blockIdx.x;
  if (fi < p.N) issue_x(frame_of(fi), 0, true);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                         // biases and the zeroed line buffer are in place

#if TSM_BNECK_STAMP
  unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  for (; fi < p.N; fi += gridDim.x) {
    const int f = frame_of(fi);
    const int fnext = fi + (int)gridDim.x < p.N ? frame_of(fi + (int)gridDim.x) : -1;
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
      BN_STAMP(6);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                 // this step's input has landed (the 16 youngest operations are the previous step's stores)
      if constexpr (IDL) __builtin_amdgcn_s_barrier();                  // ... every wave's: the identity below is read from all four slots
      BN_STAMP(0);
      // The identity operand of this step's four M-tiles, 4 x 16 bytes per lane and M-tile:
      //   CIN 256  res[mt][2 itl + qq] = bytes [64 it + 16 (2 half + qq), + 16) of the pixel's 512 (the store layout);
      //   CIN 64   res[mt][g] = channels 16 g + 8 half .. + 8 of the pixel: the B fragment of k16 group g.
      // Issue order of a step's vector-memory operations: [8 identity loads, M-tiles 0-1 (conv1 phase) | NDMA LDS-DMA of
      // the next step's input | 8 identity loads, M-tiles 2-3 (top of the conv2 phase) | 16 stores] -- every wait below counts on it.
      u32x4 res[4][4];
      // IDL: this step's M-tiles 2, 3 (row 2s) from slots 0, 1; the NEXT step's M-tiles 0, 1 (row 2s + 1) from slots 2, 3,
      // parked in accumulation registers for a step
      auto capture_res = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int k = 0; k < 4; ++k) res[mt][k] = nxt[mt][k];
#pragma unroll
        for (int ws = 0; ws < 4; ++ws)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + idrd + ws * kSlot + (k >> 1) * 2048 + ((((unsigned)(k & 1)) ^ idflip) << 4));
            if (ws < 2) res[2 + ws][k] = v;
            else nxt[ws - 2][k] = v;
          }
      };
      auto issue_res = [&](int mt) {
        if (idl) return;
        const int m = 32 * mt + l31;
        const int dr = m >= W ? 1 : 0, c = m - dr * W, r = r0 + dr;
        const bool ok = m < W2 && (unsigned)r < (unsigned)H;
        // Issued from inline asm (round 5): hipcc's wait insertion counts register loads but NOT the LDS-DMA operations queued
        // between them, so behind every counted wait below it put a vmcnt(2) / vmcnt(3) of its own in front of the first use --
        // on the hardware's counter that drains the next step's input DMA and all but three stores, once per M-tile, on every
        // wave (the uses are shared code for the waves that take the identity from LDS).  The counted waits name the registers
        // ("+v") in front of their first consumer; the ISA is audited for moves of them (tests/test_code_objects.py).
        asm volatile("s_nop 4" ::: "memory");            // (the descriptor may be fresh from scalar moves: nothing inside asm is padded)
        if constexpr (DUAL) {
#if TSM_BNECK_X & 2
          const unsigned o = (unsigned)((s * 16 + mt * 4) * 1024 + lane * 16);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(res[mt][g]) : "v"(o), "s"(rsrcR), "n"(g * 1024) : "memory");
#else
          const unsigned o = ok ? (unsigned)((r * W + c) * XROW + half * 16) : kInvalid;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(res[mt][g]) : "v"(o), "s"(rsrcR), "n"(g * 32) : "memory");
#endif
        } else {
          const unsigned o = ok ? (unsigned)((r * W + c) * 512 + (2 * wave) * 64 + 2 * half * 16) : kInvalid;
#pragma unroll
          for (int k = 0; k < 4; ++k)      // (the constant part rides in the instruction's offset: one address register per M-tile)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(res[mt][k]) : "v"(o), "s"(rsrcR), "n"((k >> 1) * 64 + (k & 1) * 16) : "memory");
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
            if constexpr (!IDL) {
              // the slot is free (its fragments are in registers): fetch the next step's input, a whole step ahead
              if (s + 1 < nsteps) issue_x(f, s + 1, true);
              else issue_x(fnext < 0 ? f : fnext, 0, fnext >= 0);
            }
          }
          // (IDL: the parked fragments of the second tile, four at a time)
#pragma unroll
          for (int g4 = 0; g4 < GH; g4 += 4) {
            u32x4 w1p[4];
            if constexpr (IDL) {
              if (gh == 1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) w1p[g] = *reinterpret_cast<const u32x4 *>(lds + kW1pOff + (g4 + g) * 1024 + lane * 16);
              }
            }
#pragma unroll
            for (int g = g4; g < g4 + 4; ++g)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                const u32x4 wf = (IDL && nt == 1 && gh == 1) ? w1p[g - g4] : w1r[nt][GH * gh + g];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf[g]), acc[nt], 0, 0, 0);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (IDL) {             // (behind the MFMAs: the slot's fragment registers are free, the reads land under the epilogue)
          if (idl) capture_res();
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
      BN_STAMP(1);
      __builtin_amdgcn_s_barrier();    // rows 2s - 2 .. 2s + 1 are complete; nobody still reads the mid tile of the previous step
      BN_STAMP(2);
      // IDL: ... nor anybody's input slot: re-arm it (a step minus the conv1 phase ahead), piece by piece behind conv2's MFMAs
      // (round 5; behind the late identity loads in the wave's issue order)
      XDma xd{};
      if constexpr (IDL) xd = (s + 1 < nsteps) ? prep_x(f, s + 1, true) : prep_x(fnext < 0 ? f : fnext, 0, fnext >= 0);
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
        // fragment reads run PXR - 1 steps ahead of their MFMA (3; the 64-channel form has registers for a ring of eight: no change)
        constexpr int PXR = TSM_BNECK_PXR64 > 4 && CIN == 64 ? TSM_BNECK_PXR64 : 4;
        u32x4 px[PXR][2];
        unsigned tb[2] = {0u, 0u};
        auto rd = [&](int st) {
          const int tap = st >> 2, g = st & 3, ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (g == 0) {
              const int pp = ((2 * s + drr[m] + ky) & 3) * kBnRP + cc[m] + kx;      // row r0 + dr - 1 + ky, column c - 1 + kx (+ 1)
              tb[m] = (unsigned)(pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4));
            }
            px[st & (PXR - 1)][m] = *reinterpret_cast<const u32x4 *>(lds + tb[m] + g * kBnT1Plane);
          }
        };
        BN_STAMP(7);      // (conv2's prologue: the late identity loads, lane geometry, accumulators)
#pragma unroll
        for (int st = 0; st < PXR - 1; ++st) rd(st);
        static_for<36>([&](auto sc) __attribute__((always_inline)) {
          constexpr int st = decltype(sc)::value;
          if constexpr (st + PXR - 1 < 36) rd(st + PXR - 1);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2r[st]), __builtin_bit_cast(bf16x8, px[st & (PXR - 1)][m]),
                                                             acc[m], 0, 0, 0);
            if constexpr (IDL && st >= 2 && st < 2 + 2 * NDMA && (st & 1) == 0) {
              if (m == 0) x_piece(xd, std::integral_constant<int, (st - 2) / 2>{});      // one piece every second step, behind an MFMA
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
        BN_STAMP(8);      // (conv2's 36 steps)
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
      BN_STAMP(3);
      __builtin_amdgcn_s_barrier();    // the mid tile is complete; the line buffer may be overwritten by the next step
      BN_STAMP(4);
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
#if TSM_BNECK_X & 1
        const unsigned yo = (unsigned)(((s * 4 + wave) * 16 + mt * 4) * 1024 + lane * 16 - (l31 * 0));   // (+ itl * 64 + qq * 16 below: patched to * 1024 there)
#else
        const unsigned yo = (ml < W2 && (unsigned)r < (unsigned)H) ? (unsigned)((r * W + c) * 512 + (2 * wave) * 64 + 2 * half * 16) : kInvalid;
#endif
        // this M-tile's identity operand; younger operations: M-tiles 0, 1 -- the other early loads (4 / 0), the next input
        // (NDMA), the late loads (8), the stores so far (0 / 4) = NDMA + 12; M-tiles 2, 3 -- the other late loads and the
        // stores so far = 12
        // (IDL waves issue no late loads: [8 early loads | NDMA | 16 stores], and M-tiles 2, 3 wait for nothing)
        // (IDL form, round 5: the input DMA rides on conv2's MFMAs, BEHIND the late loads in the issue order
        //  [8 early loads | 8 late loads | NDMA | 16 stores]: NDMA + 12 for every M-tile)
        if (!idl) {
          if constexpr (mt < 2 || IDL) wait_vmcnt(NDMA + 12);
          else wait_vmcnt(12);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(res[mt][k]));   // (no consumer of the identity is scheduled above the counted wait)
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
#if TSM_BNECK_X & 1
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)yo, (itl * 2 + qq) * 1024, TSM_AUX_BNECK);
#else
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcY, (int)yo, itl * 64 + qq * 16, TSM_AUX_BNECK);
#endif
          }
        }
      });
      BN_STAMP(5);
    }
  }
#if TSM_BNECK_STAMP
  if (blockIdx.x == 0 && lane == 0)
    printf("bneck<%d,%d,%d> wave %d: input wait %llu conv1 %llu barrier1 %llu conv2 %llu (prologue %llu steps %llu epilogue %llu) barrier2 %llu conv3 %llu between steps %llu cycles\n", CIN, (int)SHIFT,
           (int)IDL, wave, stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3] + stamp_acc[7] + stamp_acc[8], stamp_acc[7], stamp_acc[8], stamp_acc[3], stamp_acc[4], stamp_acc[5], stamp_acc[6]);
#endif
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
  if (p.cin == 256 && p.W == 64) {   // the identity from the input slots in LDS
    if (p.T > 0) TSM_KLAUNCH((bneck_ws_kernel<256, true, true>), grid, block, BnLds<256>::kBytesIdl, s, p);
    else TSM_KLAUNCH((bneck_ws_kernel<256, false, true>), grid, block, BnLds<256>::kBytesIdl, s, p);
  } else if (p.cin == 256) {
    if (p.T > 0) TSM_KLAUNCH((bneck_ws_kernel<256, true>), grid, block, BnLds<256>::kBytes, s, p);
    else TSM_KLAUNCH((bneck_ws_kernel<256, false>), grid, block, BnLds<256>::kBytes, s, p);
  } else {
    if (p.T > 0) TSM_KLAUNCH((bneck_ws_kernel<64, true>), grid, block, BnLds<64>::kBytes, s, p);
    else TSM_KLAUNCH((bneck_ws_kernel<64, false>), grid, block, BnLds<64>::kBytes, s, p);
  }
  return hipGetLastError();
}

hipError_t opt_in_bneck() {
  hipError_t first = hipSuccess;
  auto opt_in = [&](const void *fn, size_t bytes) {
    const hipError_t st = lds_opt_in(fn, bytes);
    if (st != hipSuccess && first == hipSuccess) first = st;
  };
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<256, true>), BnLds<256>::kBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<256, false>), BnLds<256>::kBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<256, true, true>), BnLds<256>::kBytesIdl);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<256, false, true>), BnLds<256>::kBytesIdl);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<64, true>), BnLds<64>::kBytes);
  opt_in(reinterpret_cast<const void *>(&bneck_ws_kernel<64, false>), BnLds<64>::kBytes);
  return first;
}

}  // namespace tsm
