// Pure-host pieces of the engine: BatchNorm folding / weight packing, the bf16 storage-format converters, the
// segment-length rule and the TSM_TUNE_CACHE line parser.  No HIP types, so this header also compiles with plain g++:
// tests/host_sanitize.cpp builds it with -fsanitize=address,undefined and fuzzes the parser (CPU only; GPU ASAN is
// not available on this pool).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace tsm_host {

constexpr float kBnEps = 1e-5f;

// Long-K fp32 layers accumulate K in segments of ~16 K-steps (512 channels-taps) so that they can also run
// split-K (one workgroup per tile and segment) with bit-identical results when the batch is too small to
// fill the chip with whole-K tiles.  The choice depends on the layer only, never on the batch size.
inline int segment_len(int kp, int prec) {
  if (prec != 0 /* kPrecF32 */) return 0;
  const int nk = kp / 32;
  if (nk < 32) return 0;
  const int nseg = nk / 16;   // (segments of 8 K-steps were measured in round 2: no gain at batch 1-4)
  return (nk + nseg - 1) / nseg;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// The tail split of a segmented 64x64 launch (ConvParams::ksplit = 2, tile code bit 0x200).  The conv_igemm SEG kernel keeps
// `wg_per_cu` (five) workgroups per CU resident, so a launch of ntm x ntn tiles runs in rounds of wg_per_cu * n_cu; when the last
// round is less than ~85 % full its tiles -- rounded DOWN to whole rows of tiles, so that the tail is a contiguous range of output
// rows -- run as (tile, K segment) pieces.  Returns the first tail tile (a multiple of ntn, in (0, ntm * ntn)), or 0 when the
// launch should stay whole-K: no whole round, no remainder worth splitting, or the segment sums [n_seg][tail rows][cout] do not
// fit `scratch_elems` floats.
inline long tail_split_point(long m_rows, int cout, int n_seg, int n_cu, size_t scratch_elems, int wg_per_cu = 5) {
  if (m_rows <= 0 || cout <= 0 || cout % 64 != 0 || n_seg < 2 || n_cu <= 0) return 0;
  const long ntn = cout / 64, ntm = (m_rows + 63) / 64, tiles = ntm * ntn, slots = (long)wg_per_cu * n_cu;
  const long rounds = tiles / slots, rem = tiles - rounds * slots;
  if (rounds < 1 || rem == 0 || rem * 100 > slots * 85) return 0;
  const long from = rounds * slots / ntn * ntn;
  if (from <= 0 || from >= tiles) return 0;
  const size_t tail_rows = (size_t)(m_rows - (from / ntn) * 64);
  if ((size_t)n_seg * tail_rows * (size_t)cout > scratch_elems) return 0;
  return from;
}

inline uint16_t f2bf(float f) {  // round to nearest even, like v_cvt_pk_bf16_f32
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// fp32 -> bf16, two elements per float slot (the vector shrinks to half its length).
inline void to_bf16(std::vector<float> *v) {
  std::vector<float> out((v->size() + 1) / 2, 0.f);
  uint16_t *o = reinterpret_cast<uint16_t *>(out.data());
  for (size_t i = 0; i < v->size(); ++i) o[i] = f2bf((*v)[i]);
  v->swap(out);
}
// In place: every group of 8 consecutive floats becomes [hi x8 | lo x8] bf16 (32 bytes, same size).
inline void to_split(std::vector<float> *v) {
  uint16_t g[16];
  for (size_t i = 0; i + 8 <= v->size(); i += 8) {
    for (int e = 0; e < 8; ++e) {
      const float x = (*v)[i + e];
      g[e] = f2bf(x);
      g[8 + e] = f2bf(x - bf2f(g[e]));
    }
    std::memcpy(v->data() + i, g, 32);
  }
}

// Stem weights for the bf16 formats, whose input is stored as pixel pairs (tsm_ops.hip, pack_input_kernel):
// K = (ky, pair j, pixel-in-pair q, c4) = 7 x 4 x 2 x 4 = 224, covering pixels 2ox-4 .. 2ox+3, i.e. kx = 2j + q - 1
// (kx = -1 and c = 3 carry zero weights).
inline void fold_and_pack_stem_pairs(const float *w, const float *gamma, const float *beta, const float *mean,
                              const float *var, int cout, int kp, std::vector<float> *wp, std::vector<float> *bias) {
  wp->assign((size_t)cout * kp, 0.f);
  bias->resize(cout);
  for (int o = 0; o < cout; ++o) {
    const float scale = gamma[o] / std::sqrt(var[o] + kBnEps);
    (*bias)[o] = beta[o] - mean[o] * scale;
    float *dst = wp->data() + (size_t)o * kp;
    for (int c = 0; c < 3; ++c)
      for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) {
          const int j = (kx + 1) >> 1, q = (kx + 1) & 1;
          dst[((ky * 4 + j) * 2 + q) * 4 + c] = w[(((size_t)o * 3 + c) * 7 + ky) * 7 + kx] * scale;
        }
  }
}

// Fold BN into the conv and pack OIHW -> [Cout][Kp], K = (ky, kx, c) with c padded to cp.
inline void fold_and_pack(const float *w, const float *gamma, const float *beta, const float *mean,
                   const float *var, int cout, int cin, int k, int cp, int kp, std::vector<float> *wp,
                   std::vector<float> *bias) {
  wp->assign((size_t)cout * kp, 0.f);
  bias->resize(cout);
  for (int o = 0; o < cout; ++o) {
    const float scale = gamma[o] / std::sqrt(var[o] + kBnEps);
    (*bias)[o] = beta[o] - mean[o] * scale;
    float *dst = wp->data() + (size_t)o * kp;
    for (int c = 0; c < cin; ++c)
      for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx)
          dst[(ky * k + kx) * cp + c] = w[(((size_t)o * cin + c) * k + ky) * k + kx] * scale;
  }
}

// Tuned tile shapes are cached per power-of-two bucket of the clip count (ragged last batches of a video
// would otherwise each pay a tuning pass): the first clip count that lands in a bucket tunes it.
inline int tile_bucket(int n_clips) {
  int b = 1;
  while (b < n_clips) b <<= 1;
  return b;
}


// conv3 weights of a fused conv2+conv3 block (conv23_fused_kernel) in MFMA-fragment order.  w3p is the packed,
// BN-folded matrix [4 * cmid][cmid]; the kernel's wave `wn` (of cmid / 32) loads, for output chunk j (of 4, cmid
// channels each) and k-group kk (of cmid / 8), ONE 16-byte element per lane:
//   out[(((j * wgn + wn) * nkk + kk) * 64 + lane) * 4 + s] = W3[n = j * cmid + wn * 32 + (lane & 31)][k = 8 kk + 4 (lane >> 5) + s]
// i.e. exactly the B operand of v_mfma_f32_32x32x2_f32 step s of that k-group, so the load is lane-linear (coalesced).
inline void pack_w3_fragments(const float *w3p, int cmid, std::vector<float> *out) {
  const int wgn = cmid / 32, nkk = cmid / 8;
  out->assign((size_t)4 * cmid * cmid, 0.f);
  for (int j = 0; j < 4; ++j)
    for (int wn = 0; wn < wgn; ++wn)
      for (int kk = 0; kk < nkk; ++kk)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int n = j * cmid + wn * 32 + (lane & 31), k = 8 * kk + 4 * (lane >> 5) + s;
            (*out)[((((size_t)j * wgn + wn) * nkk + kk) * 64 + lane) * 4 + s] = w3p[(size_t)n * cmid + k];
          }
}

// The same for a split-bf16 engine: per output chunk j, wave wn and k16 group kq the kernel loads TWO 16-byte elements
// per lane, the hi and the lo halves of the 8 channels k = 16 kq + 8 (lane >> 5) + 0..7 of row n:
//   element (((j * wgn + wn) * (2 * nkq) + 2 * kq + part) * 64 + lane), part 0 = hi x8, 1 = lo x8  (bf16 pairs per float slot)
// with hi = bf16(w), lo = bf16(w - hi) exactly as to_split() stores the conv3 weights of the un-fused path.
inline void pack_w3_fragments_split(const float *w3p, int cmid, std::vector<float> *out) {
  const int wgn = cmid / 32, nkq = cmid / 16;
  out->assign((size_t)4 * cmid * cmid, 0.f);
  uint16_t *o = reinterpret_cast<uint16_t *>(out->data());
  for (int j = 0; j < 4; ++j)
    for (int wn = 0; wn < wgn; ++wn)
      for (int kq = 0; kq < nkq; ++kq)
        for (int lane = 0; lane < 64; ++lane) {
          const int n = j * cmid + wn * 32 + (lane & 31), k0 = 16 * kq + 8 * (lane >> 5);
          const size_t base = ((((size_t)j * wgn + wn) * (2 * nkq) + 2 * kq) * 64 + lane) * 8;   // in bf16 units
          for (int e = 0; e < 8; ++e) {
            const float x = w3p[(size_t)n * cmid + k0 + e];
            const uint16_t hi = f2bf(x);
            o[base + e] = hi;
            o[base + 64 * 8 + e] = f2bf(x - bf2f(hi));
          }
        }
}

// One line of a TSM_TUNE_CACHE file: "<signature>|<bucket>|c0,c1,...".  Succeeds only when the line starts with
// `want`, holds exactly codes->size() integers and each is a ConvTile below `num_tiles`, optionally | 0x100 (split-K) | 0x400 (block runs conv2 + conv3 fused) | 0x800 (the whole block runs as one launch) | 0x1000 (conv3 also runs the next block's conv1) | 0x2000 (conv1 also runs the block's stride-2 conv2).
// Anything else (foreign keys, truncated lines, garbage, overlong numbers) leaves *codes untouched.
inline bool parse_tune_line(const char *line, const std::string &want, int num_tiles, std::vector<int> *codes) {
  if (strncmp(line, want.c_str(), want.size()) != 0) return false;
  std::vector<int> got;
  const char *q = line + want.size();
  while (*q && *q != '\n') {
    char *end = nullptr;
    const long v = strtol(q, &end, 10);
    if (end == q) return false;
    if (v < 0 || (v & ~0x3F0FL) != 0 || (int)(v & 15) >= num_tiles) return false;
    got.push_back((int)v);
    if (*end == ',') q = end + 1;
    else if (*end == '\n' || *end == 0) q = end;
    else return false;
  }
  if (got.size() != codes->size()) return false;
  *codes = got;
  return true;
}

}  // namespace tsm_host
