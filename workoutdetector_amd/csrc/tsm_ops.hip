// Small streaming kernels around the conv stack, and device_info().
#include "tsm_device.h"

#include <string>

namespace tsm {

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


#define TSM_DISPATCH_FMT(prec, KERNEL, grid, stream, ...)                                                   \
  do {                                                                                                      \
    if ((prec) == kPrecBf16x3)                                                                              \
      TSM_KLAUNCH((KERNEL<kPrecBf16x3>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);             \
    else if ((prec) == kPrecBf16)                                                                           \
      TSM_KLAUNCH((KERNEL<kPrecBf16>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);               \
    else                                                                                                    \
      TSM_KLAUNCH((KERNEL<kPrecF32>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                \
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
    TSM_KLAUNCH(from_f32_kernel<kPrecBf16x3>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else if (prec == kPrecBf16)
    TSM_KLAUNCH(from_f32_kernel<kPrecBf16>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_to_f32(const float *x, float *y, int64_t n8, int prec, hipStream_t s) {
  if (prec == kPrecBf16x3)
    TSM_KLAUNCH(to_f32_kernel<kPrecBf16x3>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
  else if (prec == kPrecBf16)
    TSM_KLAUNCH(to_f32_kernel<kPrecBf16>, dim3(grid_for(n8, 8192)), dim3(256), 0, s, x, y, n8);
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
    TSM_KLAUNCH(preprocess_kernel<unsigned char>, dim3(grid), dim3(256), 0, s, p);
  else
    TSM_KLAUNCH(preprocess_kernel<float>, dim3(grid), dim3(256), 0, s, p);
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
    TSM_KLAUNCH(gather_clips_kernel, dim3(gx, (unsigned)ny), dim3(256), 0, s, p);
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
  TSM_KLAUNCH(temporal_shift_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, x, y, n_frames,
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
    TSM_KLAUNCH(head_pool_kernel<kPrecBf16x3>, grid, dim3(256), 0, s, feat, pooled, hw, c);
  else if (prec == kPrecBf16)
    TSM_KLAUNCH(head_pool_kernel<kPrecBf16>, grid, dim3(256), 0, s, feat, pooled, hw, c);
  else
    TSM_KLAUNCH(head_pool_kernel<kPrecF32>, grid, dim3(256), 0, s, feat, pooled, hw, c);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  TSM_KLAUNCH(head_fc_kernel, dim3(n_clips, num_class), dim3(64), 0, s, pooled, fc_w, fc_b, logits, c,
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
  TSM_KLAUNCH(scores_to_states_kernel, dim3((n + 63) / 64), dim3(64), 0, s, logits, n, c, softmax, threshold, states, top);
  return hipGetLastError();
}


// ---- launch trace (tests): which kernels did this thread launch? ---------------------------------------------------
namespace {
struct LaunchTrace {
  bool on = false;
  std::string text;
};
thread_local LaunchTrace g_trace;
}  // namespace

void note_launch(const char *kernel, const char *where) {
  if (!g_trace.on) return;
  // "(conv_igemm<BM, BN, ...>)" as the launch site spells it; inside a template the enclosing function's
  // "[BM = 64, BN = 64, ...]" (clang's __PRETTY_FUNCTION__) resolves the names
  std::string k(kernel);
  if (!k.empty() && k.front() == '(' && k.back() == ')') k = k.substr(1, k.size() - 2);
  const char *with = where ? strstr(where, " [") : nullptr;
  g_trace.text += k;
  if (with) g_trace.text += with;
  g_trace.text += '\n';
}
void trace_launches(bool on) {
  g_trace.on = on;
  if (on) g_trace.text.clear();
}
const char *launch_trace() { return g_trace.text.c_str(); }

hipError_t lds_opt_in(const void *kernel, size_t bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

const DeviceInfo &device_info() {
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
  for (hipError_t st : {opt_in_bf16_256(), opt_in_ws(), opt_in_bneck(), opt_in_conv31(), opt_in_front()})
    if (st != hipSuccess && di.status == hipSuccess) di.status = st;
  return seen.emplace(dev, di).first->second;
}

}  // namespace tsm
