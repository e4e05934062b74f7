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
//     row r: [8 LDS-DMA pieces of row r + 3] [r odd: 2 stores]
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

template <bool SHIFT>
__global__ void __launch_bounds__(256, 1) front_s2_kernel(const FrontParams p) {
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
  // LDS-DMA of row r of frame f into row slot `slot`: always 8 operations (dead ones fetch nothing)
  auto issue_row = [&](int f, int r, int slot, bool live) {
    const int tt = p.T > 0 ? f % p.T : 0;
    // the descriptor starts one frame BEFORE f (only ever addressed there when frame t - 1 exists)
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x) + ((long)f - 1) * xframe), 0, 3 * xframe, 0x00020000);
    unsigned char *dst = lds + kFrXOff + slot * kFrSlot + 4 * wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 32 * j + (lane >> 1);
      const bool okp = live && c < W;
      const unsigned own = (unsigned)xframe + (unsigned)((r * W + c) * 512 + 4 * wave * 32 + hsel * 16);
      const unsigned vC = okp ? own : kInvalid;
      const unsigned vA = (okp && tt < p.T - 1) ? own + (unsigned)xframe : kInvalid;
      const unsigned vB = (okp && tt > 0) ? own - (unsigned)xframe : kInvalid;
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        // fold = 32 channels: planes 0, 1 from frame t + 1, planes 2, 3 from t - 1 (all of them wave 0's)
        const unsigned v = (SHIFT && wave == 0) ? (pl < 2 ? vA : vB) : vC;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_void *)(dst + pl * 2048 + j * 1024), 16, (int)v, pl * 32, 0, 0);
      }
    }
  };

  // the (frame, row) the loader is at: three rows ahead of the row being multiplied
  int fi = blockIdx.x;                   // virtual index of the frame being multiplied
  int lfi = fi, lrow = 0;                // ... of the frame / row being requested
  auto issue_next = [&](int slot) {
    issue_row(lfi < p.N ? frame_of(lfi) : 0, lrow, slot, lfi < p.N);
    if (++lrow == H) {
      lrow = 0;
      lfi += (int)gridDim.x;
    }
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
    for (int r = 0; r < H; r += 2) {
#pragma unroll
      for (int odd = 0; odd < 2; ++odd) {
        // ================= conv1: row r + odd -> line-buffer slot lslot =================
        // this wave's pieces of the row have landed; younger: the pieces of the two rows behind it (16) and the stores of the odd
        // rows among the last three (r even: two of them, r odd: one)
        if (odd == 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __builtin_amdgcn_s_barrier();    // ... every wave's; nobody still reads the line buffer (conv2 of the previous row pair is over)
        if (r == 0 && odd == 0) {        // row -1 of the new frame: zeros in slot 0 (the previous frame's row H - 2 lived there)
          for (int i = tid; i < 8 * kFrRP * 2; i += 256) {
            const int pl = i / (kFrRP * 2), q = i - pl * (kFrRP * 2);
            *reinterpret_cast<u32x4 *>(lds + pl * kFrLbPlane + q * 16) = u32x4{0u, 0u, 0u, 0u};
          }
        }
        {
          const unsigned char *xs = lds + kFrXOff + xslot * kFrSlot;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            u32x4 xf[8];                   // the pixel's fragments, eight k16 groups at a time
#pragma unroll
            for (int gh = 0; gh < 2; ++gh) {
#pragma unroll
              for (int g = 0; g < 8; ++g) xf[g] = *reinterpret_cast<const u32x4 *>(xs + xrd[mt] + (8 * gh + g) * 2048);
#pragma unroll
              for (int g = 0; g < 8; ++g)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[8 * gh + g]), __builtin_bit_cast(bf16x8, xf[g]), acc, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
            // bias1, ReLU, bf16; the swap pairs groups (0, 1) and (2, 3): this lane then holds channels 16 g' + 8 half .. + 8 of
            // k16 group g' = 2 wave + qq of its pixel = one 16-byte half of the pixel's entry in plane g' of the line buffer
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
                *reinterpret_cast<u32x4 *>(lds + (2 * wave + qq) * kFrLbPlane + lslot * (kFrRP * 32) + lbw[mt]) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();    // the line-buffer row is complete; every wave has read input slot xslot: re-arm it, three rows ahead
        issue_next(xslot);
        if (odd == 1) {
          // ================= conv2: output row s = r / 2 from line-buffer rows r - 1, r, r + 1 =================
          const int s = r >> 1;
          // (slot of row r - 1 + ky: lslot is row r + 1's, i.e. ky = 2; ky = 1 -> lslot - 1, ky = 0 -> lslot - 2, mod 3)
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          u32x4 px[4];
          unsigned rb[3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            int sl = lslot + ky - 2;
            sl += sl < 0 ? 3 : 0;
            rb[ky] = (unsigned)(sl * (kFrRP * 32));
          }
          auto rd = [&](int st) {
            const int tap = st >> 3, g = st & 7, ky = tap / 3, kx = tap - ky * 3;
            px[st & 3] = *reinterpret_cast<const u32x4 *>(lds + rb[ky] + lbr[kx] + g * kFrLbPlane);
          };
          rd(0); rd(1); rd(2);
          static_for<72>([&](auto sc) __attribute__((always_inline)) {
            constexpr int st = decltype(sc)::value;
            if constexpr (st + 3 < 72) rd(st + 3);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2r[st]), __builtin_bit_cast(bf16x8, px[st & 3]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          });
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
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dead rows of the tail land before the workgroup leaves its LDS
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
