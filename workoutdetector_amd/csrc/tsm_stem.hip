// The 7x7 stride-2 stem as a direct convolution, with the 3x3 max-pool fused behind it.
#include "tsm_device.h"

namespace tsm {

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

// PLANAR: x is the caller's [N, 3, H, W] fp32 tensor itself (the reference model's input layout) instead of the packed pairs
// pack_input_kernel would have written: a patch chunk is then six floats (two pixels x three colour planes) that are held
// raw while the previous tile computes and rounded / split exactly as pack_input does (store_group's arithmetic) when the
// patch is written to LDS -- same bits, and the forward loses the pack launch with its write and re-read of the packed tensor.
#ifndef TSM_STEM_WREG
#define TSM_STEM_WREG 4
#endif
#ifndef TSM_STEM_STAMP
#define TSM_STEM_STAMP 0   // diagnostic builds only: per-phase cycle sums of workgroup 0 (s_memtime), printed at the kernel's end
#endif
#if TSM_STEM_STAMP
#define ST_STAMP(i)                                         \
  do {                                                      \
    const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += _t - stamp_last;                        \
    stamp_last = _t;                                        \
  } while (0)
#else
#define ST_STAMP(i) do {} while (0)
#endif
template <bool X3, bool PLANAR>
// (bf16: 77 760 B of LDS and <= 128 registers, so that TWO workgroups share a CU and overlap each other's phases)
__global__ void __launch_bounds__(512, X3 ? 1 : 4) stem_pool_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ y, int n,
                                                        int hi, int wi, int ho, int wo, int hp, int wp, int kp, int relu) {
  constexpr int NT = 512;
  constexpr int GB = X3 ? 32 : 16;
  constexpr int WROW = X3 ? 1040 : 464;   // weight row stride: data + padding, conflict-free ds_read_b128 over 16 rows
  constexpr int CSB = X3 ? 256 * 68 * 4 : 256 * 72 * 2;   // the conv tile: [256][68] fp32, or (bf16) [256][72] bf16 bit patterns
  constexpr int OPX = X3 ? 256 : 128;
  // The patch in LDS.  bf16: input row R starts at 16-byte slot 24 R + (R >> 1) -- pitch 24 instead of the 20 pairs a row
  // holds, and one slot more every second row -- so that conv pixel mi = 17 mr + mc reads its A fragment from slot
  // 49 mr + mc + const == mi + const (mod 16): the 16 lanes of a ds_read_b128 group (lanes {0-3, 12-15, 20-27} / {4-11, 16-19,
  // 28-31}: MI355X_MICROARCH.md, LDS) are 16 different residues of mi, i.e. 16 different bank slots, ALSO where a wave's 32
  // pixels wrap around the 17-pixel rows of the conv tile.  (Pitch 20: every row wrap shifts the slots by 40 - 17 = 7 mod 16,
  // half of the A reads were 2-way conflicted -- the model of tools/probes says 51 % of their LDS cycles, rocprofv3 39 % of the
  // kernel's.)  Split-bf16 keeps the dense rows (32-byte groups; one workgroup per CU there).
  constexpr int PPITCH = X3 ? kPoolPC : 24;
  constexpr int PSLOTS = X3 ? kPoolPR * kPoolPC : kPoolPR * 24 + (kPoolPR >> 1) + 1;     // groups of the patch image
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * WROW + PSLOTS * GB + CSB];
  unsigned char *Ws = smem;
  unsigned char *Ps = smem + 64 * WROW;
  float *Cs = reinterpret_cast<float *>(smem + 64 * WROW + PSLOTS * GB);  // X3: [256][68] fp32
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
  const unsigned char *a_base = Ps + ((2 * mr) * PPITCH + (X3 ? 0 : mr) + mc) * GB;     // (input row 2 mr + ky: + ky * PPITCH + (ky >> 1) below)
  const unsigned char *b_base = Ws + l31 * WROW;

  constexpr int PCH = kPoolPR * kPoolPC * GB / 16;
  constexpr int PPASS = (PCH + NT - 1) / NT;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  // where chunk ci of the patch (row-major over its 35 x 20 groups) lives in LDS
  auto patch_slot = [&](int ci) -> int {
    if constexpr (X3) return ci * 16;
    const int r = ci / kPoolPC, c = ci - r * kPoolPC;
    return (r * PPITCH + (r >> 1) + c) * 16;
  };
  u32x4 pre[PLANAR ? 1 : PPASS];
  u32x2_t raw[PLANAR ? PPASS : 1][3];               // PLANAR: (pixel 2j, pixel 2j + 1) of each colour plane, fp32 bits
  const unsigned plane_bytes = (unsigned)hi * wi * 4;
  const bool even_w = (wi & 1) == 0;                // pairs never straddle a row end and are 8-byte aligned: one load per plane
  auto fetch_patch = [&](int t) {
    const int f = t / tiles_f, rem = t - f * tiles_f, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    // conv tile origin (2*py0 - 1, 2*px0 - 1)  ->  input rows from 2*(2*py0 - 1) - 3, pairs from (2*px0 - 1) - 2
    const int iy0 = 4 * ty * kPoolPH - 5, pc0 = 2 * tx * kPoolPW - 3;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(x) + (size_t)f * (PLANAR ? 3 * (size_t)plane_bytes : (size_t)x_frame)), 0,
        (int)(PLANAR ? 3 * plane_bytes : x_frame), 0x00020000);
#pragma unroll
    for (int q = 0; q < PPASS; ++q) {
      const int ci = tid + q * NT;
      const int g = X3 ? ci >> 1 : ci;
      const int r = g / kPoolPC, c = g - r * kPoolPC;
      const int iy = iy0 + r, pcx = pc0 + c;
      const bool ok = ci < PCH && (unsigned)iy < (unsigned)hi && (unsigned)pcx < (unsigned)wpairs;
      if constexpr (PLANAR) {
        const unsigned off = (unsigned)((iy * wi + 2 * pcx) * 4);
        const bool ok1 = ok && 2 * pcx + 1 < wi;
        if (even_w) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            raw[q][pl] = __builtin_amdgcn_raw_buffer_load_b64(rsrcX, (int)(ok ? off : kInvalid), (int)(pl * plane_bytes), 0);
        } else {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            raw[q][pl][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrcX, (int)(ok ? off : kInvalid), (int)(pl * plane_bytes), 0);
            raw[q][pl][1] = __builtin_amdgcn_raw_buffer_load_b32(rsrcX, (int)(ok1 ? off + 4u : kInvalid), (int)(pl * plane_bytes), 0);
          }
        }
      } else {
        const unsigned off = (unsigned)((iy * wpairs + pcx) * GB + (X3 ? (ci & 1) * 16 : 0));
        pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)(ok ? off : kInvalid), 0, 0);
      }
    }
  };
  // PLANAR: the chunk pack_input_kernel would have stored for this thread's pair (store_group<bf16 / split-bf16>)
  auto packed_chunk = [&](int q) -> u32x4 {
    if constexpr (PLANAR) {
      // (element -> scalar first: __builtin_bit_cast applied to a vector-ELEMENT expression reads element 0 whatever the
      //  index with this hipcc -- the second pixel's loads were dropped and the first pixel's words stored twice)
      const unsigned u0a = raw[q][0][0], u0b = raw[q][0][1], u1a = raw[q][1][0], u1b = raw[q][1][1], u2a = raw[q][2][0], u2b = raw[q][2][1];
      const float c0a = __builtin_bit_cast(float, u0a), c0b = __builtin_bit_cast(float, u0b);
      const float c1a = __builtin_bit_cast(float, u1a), c1b = __builtin_bit_cast(float, u1b);
      const float c2a = __builtin_bit_cast(float, u2a), c2b = __builtin_bit_cast(float, u2b);
      if constexpr (X3) {
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        split_pair(c0a, c1a, &h0, &l0);
        split_pair(c2a, 0.f, &h1, &l1);
        split_pair(c0b, c1b, &h2, &l2);
        split_pair(c2b, 0.f, &h3, &l3);
        return ((tid + q * NT) & 1) ? u32x4{l0, l1, l2, l3} : u32x4{h0, h1, h2, h3};
      } else {
        return u32x4{pack_bf16(c0a, c1a), pack_bf16(c2a, 0.f), pack_bf16(c0b, c1b), pack_bf16(c2b, 0.f)};
      }
    } else {
      return pre[q];
    }
  };
  // (tiles in XCD-chunked order: horizontally and vertically neighbouring tiles share up to half of their patch lines)
#if TSM_STEM_STAMP
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  if ((int)blockIdx.x < n_tiles) fetch_patch((int)xcd_chunked(blockIdx.x, n_tiles));
  // bf16: the weight fragments of this lane's second channel tile (14 x 16 bytes) live in registers for the life of the workgroup
  // -- the kernel used 56 of the 128 registers two workgroups per CU leave it, and every wave read the same 28 fragments from
  // LDS for every tile: one LDS read in three of the MFMA phase is gone (TSM_STEM_WREG)
  constexpr int NWREG = X3 ? 0 : TSM_STEM_WREG;     // fragments held (the first NWREG k16 steps)
  constexpr bool kWreg = NWREG > 0;
  u32x4 wreg[kWreg ? NWREG : 1];
  if constexpr (kWreg) {
    __syncthreads();      // (the weights are in LDS)
#pragma unroll
    for (int s16 = 0; s16 < NWREG; ++s16) wreg[s16] = *reinterpret_cast<const u32x4 *>(b_base + (2 * s16 + half) * GB + 32 * WROW);
  }
  for (int v = blockIdx.x; v < n_tiles; v += gridDim.x) {
    const int t = (int)xcd_chunked(v, n_tiles);
    const int f = t / tiles_f, rem = t - f * tiles_f, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int py0 = ty * kPoolPH, px0 = tx * kPoolPW;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;   // conv pixel of tile position (0, 0)
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(y) + (size_t)f * y_frame, 0, (int)y_frame, 0x00020000);
    ST_STAMP(5);
    __syncthreads();
    ST_STAMP(0);
#pragma unroll
    for (int q = 0; q < PPASS; ++q)
      if (tid + q * NT < PCH) *reinterpret_cast<u32x4 *>(Ps + patch_slot(tid + q * NT)) = packed_chunk(q);
    if (v + (int)gridDim.x < n_tiles) fetch_patch((int)xcd_chunked(v + gridDim.x, n_tiles));
    ST_STAMP(1);
    __syncthreads();
    ST_STAMP(2);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int s16 = 0; s16 < 14; ++s16) {
      const int g = 2 * s16 + half;
      const unsigned char *ap = a_base + ((g >> 2) * PPITCH + (X3 ? 0 : (g >> 3)) + (g & 3)) * GB;     // ky = g >> 2: + (ky >> 1) slots in the bf16 image
      const unsigned char *bp = b_base + g * GB;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap));
      const bf16x8 bh0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bp));
      const bf16x8 bh1 = __builtin_bit_cast(bf16x8, (kWreg && s16 < NWREG) ? wreg[(kWreg && s16 < NWREG) ? s16 : 0] : *reinterpret_cast<const u32x4 *>(bp + 32 * WROW));
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
    ST_STAMP(3);
    __syncthreads();
    ST_STAMP(4);
    if (tid < kPoolPH * kPoolPW * 8) {  // one 8-channel group of one pooled pixel per thread
      const int pp = tid >> 3, cg = tid & 7;
      const int pyl = pp / kPoolPW, pxl = pp - pyl * kPoolPW;
      const int py = py0 + pyl, px = px0 + pxl;
      const bool ok = py < hp && px < wp;
      const unsigned base = ok ? (unsigned)((py * wp + px) * OPX + cg * GB) : kInvalid;
      if constexpr (!X3) {
        if (relu) {
          // Behind a ReLU the tile holds bf16 bit patterns of non-negative numbers and 0xFF80 (-inf, outside the image): for
          // those the 16-bit SIGNED integer order is the float order, so the 3x3 maximum is 4 v_pk_max_i16 per window
          // position on the patterns themselves -- no unpacking, no re-packing (the float path below: ~150 instructions
          // per 8 channels, on a kernel bound by its instruction count).  A NaN cannot be in the tile (fmaxf(NaN, 0) = 0
          // in the conv epilogue); without the ReLU the values may be negative and the float path runs.
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 mi = {(short)0xFF80, (short)0xFF80, (short)0xFF80, (short)0xFF80, (short)0xFF80, (short)0xFF80, (short)0xFF80, (short)0xFF80};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const u32x4 pkd = *reinterpret_cast<const u32x4 *>(Cs16 + ((2 * pyl + ky) * kPoolCC + 2 * pxl + kx) * 72 + cg * 8);
              mi = __builtin_elementwise_max(mi, __builtin_bit_cast(s16x8, pkd));
            }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mi), rsrcY, (int)base, 0, 0);
          continue;
        }
      }
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
#if TSM_STEM_STAMP
  if (blockIdx.x == 0 && (tid == 0 || tid == 448))
    printf("stem<%d,%d> wave %d: barrier1 %llu stage patch %llu barrier2 %llu mfma + tile epilogue %llu barrier3 %llu pool + store %llu cycles\n", (int)X3, (int)PLANAR,
           tid >> 6, stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3], stamp_acc[4], stamp_acc[5]);
#endif
}

// fp32 form of stem_pool.  Input NHWC4 (one 16-byte group per pixel), weights [64][Kp] fp32 with K = (ky, kx, c4);
// the conv tile's patch is 35 rows x 39 pixels.  MFMA sequence = conv_igemm's fp32 stem exactly: v_mfma_f32_32x32x2_f32
// sums k = {4 * tap + s of lane-half 0, of lane-half 1}, taps taken in pairs (2g, 2g + 1), s = 0..3, g = 0..24, so
// the results are bit-identical to it (tap 49 is K padding: zero weights, its A operand re-reads tap 48).
constexpr int kPoolPCF = 2 * (kPoolCC - 1) + 7;   // 39 input pixels per patch row

template <bool PLANAR>   // (PLANAR: x = [N, 3, H, W] fp32, see stem_pool_kernel)
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
  u32x4 pre[PPASS];                                // (PLANAR: words 0-2 loaded from the three colour planes, word 3 = 0)
  const unsigned plane_bytes = (unsigned)hi * wi * 4;
  auto fetch_patch = [&](long t) {
    const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), f = (int)(t / ((long)tiles_x * tiles_y));
    const int iy0 = 4 * ty * kPoolPH - 5, ix0 = 4 * tx * kPoolPW - 5;   // 2 * (2 * p0 - 1) - 3
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(x) + (size_t)f * (PLANAR ? 3 * (size_t)plane_bytes : (size_t)x_frame)), 0,
        (int)(PLANAR ? 3 * plane_bytes : x_frame), 0x00020000);
#pragma unroll
    for (int q = 0; q < PPASS; ++q) {
      const int ci = tid + q * NT;
      const int r = ci / kPoolPCF, c = ci - r * kPoolPCF;
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = ci < PCH && (unsigned)iy < (unsigned)hi && (unsigned)ix < (unsigned)wi;
      if constexpr (PLANAR) {
        const unsigned off = ok ? (unsigned)((iy * wi + ix) * 4) : kInvalid;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) pre[q][pl] = __builtin_amdgcn_raw_buffer_load_b32(rsrcX, (int)off, (int)(pl * plane_bytes), 0);
        pre[q][3] = 0u;
      } else {
        pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)(ok ? (unsigned)((iy * wi + ix) * 16) : kInvalid), 0, 0);
      }
    }
  };
  if ((long)blockIdx.x < n_tiles) fetch_patch(xcd_chunked(blockIdx.x, n_tiles));
  for (long v = blockIdx.x; v < n_tiles; v += gridDim.x) {
    const long t = xcd_chunked(v, n_tiles);   // (see stem_pool_kernel)
    const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), f = (int)(t / ((long)tiles_x * tiles_y));
    const int py0 = ty * kPoolPH, px0 = tx * kPoolPW;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(y) + (size_t)f * y_frame, 0, (int)y_frame, 0x00020000);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PPASS; ++q)
      if (tid + q * NT < PCH) *reinterpret_cast<u32x4 *>(Ps + (tid + q * NT) * 16) = pre[q];
    if (v + gridDim.x < n_tiles) fetch_patch(xcd_chunked(v + gridDim.x, n_tiles));
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
                            int relu, int prec, hipStream_t s, int planar) {
  const int ho = (hi + 6 - 7) / 2 + 1, wo = (wi + 6 - 7) / 2 + 1;
  const int hp = (ho + 2 - 3) / 2 + 1, wp = (wo + 2 - 3) / 2 + 1;
  if (!x || !w || !bias || !y || n <= 0 || hi <= 0 || wi <= 0 || kp < 200) return hipErrorInvalidValue;
  if (prec != kPrecBf16 && prec != kPrecBf16x3 && prec != kPrecF32) return hipErrorInvalidValue;
  if ((double)hi * wi * 16.0 > 2.0e9 || (prec != kPrecF32 && kp < 224)) return hipErrorInvalidValue;
  const long tiles = (long)n * ((hp + kPoolPH - 1) / kPoolPH) * ((wp + kPoolPW - 1) / kPoolPW);
  if (tiles >= (1L << 31) - 1024) return hipErrorInvalidValue;
  const long cap = (long)device_info().n_cu * (prec == kPrecBf16 ? 2 : 1);   // persistent: one 8-wave workgroup per CU (bf16: two)
  const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
#define TSM_STEM_ARGS dim3(grid), dim3(512), 0, s, x, w, bias, y, n, hi, wi, ho, wo, hp, wp, kp, relu
  if (prec == kPrecF32) {
    if (planar) TSM_KLAUNCH(stem_pool_f32_kernel<true>, TSM_STEM_ARGS);
    else TSM_KLAUNCH(stem_pool_f32_kernel<false>, TSM_STEM_ARGS);
  } else if (prec == kPrecBf16) {
    if (planar) TSM_KLAUNCH((stem_pool_kernel<false, true>), TSM_STEM_ARGS);
    else TSM_KLAUNCH((stem_pool_kernel<false, false>), TSM_STEM_ARGS);
  } else {
    if (planar) TSM_KLAUNCH((stem_pool_kernel<true, true>), TSM_STEM_ARGS);
    else TSM_KLAUNCH((stem_pool_kernel<true, false>), TSM_STEM_ARGS);
  }
#undef TSM_STEM_ARGS
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
    TSM_KLAUNCH((stem_direct_kernel<false, 4>), dim3(grid), dim3(256), 0, s, x, w, bias, y, n, hi, wi, ho, wo, kp, relu);
  else
    TSM_KLAUNCH((stem_direct_kernel<true, 8>), dim3(grid), dim3(512), 0, s, x, w, bias, y, n, hi, wi, ho, wo, kp, relu);
  return hipGetLastError();
}

}  // namespace tsm
