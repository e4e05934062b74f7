// Internal launch API shared by the kernel files (csrc/tsm_*.hip, device code; tsm_device.h lists them) and tsm_engine.hip (host engine).
// gfx950 only.  All activations NHWC fp32; weights packed [Cout][Kp] with K = (ky, kx, c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsm {

struct ConvParams {
  const float *x;     // [N, Hi, Wi, C]   C = 4 (stem, padded) or a multiple of 32
  const float *w;     // [Cout][Kp]       BN scale folded in, zero padded to Kp
  const float *bias;  // [Cout]           folded BN bias
  const float *res;   // nullable [M, Cout] residual added before the activation
  float *y;           // [M, Cout]
  int N, Hi, Wi, C, logC4;  // (C / 4) == 1 << logC4
  int Ho, Wo, Cout;
  int stride, pad;
  int Kp;    // padded K, multiple of 32
  int M;     // N * Ho * Wo
  int relu;
  int T;     // > 0: temporal shift over T segments fused into the A loader (1x1, stride 1)
  int fold;  // C / shift_div
  int ntm, ntn;
  int tile;  // 0 = heuristic, else a ConvTile chosen by the engine's autotuner
  int prec;  // ConvPrec: storage format of x, w, res, y and the MFMA used
  // Second A source, concatenated along K behind the first (1x1 convs only): fuses
  //   y = act( conv1x1(x, W) + conv1x1_strided(x2, W2) + bias )
  // i.e. Bottleneck.conv3 + the downsample branch of a stage's first block in one GEMM, so the
  // identity tensor is never written or read.  x2 == nullptr: single source.
  const float *x2;
  int C2, Hi2, Wi2, stride2;
  int K1;    // channels of the first source (its K extent); Kp = K1 + C2
  // Segmented K accumulation (fp32, 64x64 / 32x32 tiles): kseg_len > 0 sums K in consecutive segments of kseg_len
  // K-steps, each from a zero accumulator, and adds the segment sums in order:  out = ((0 + s0) + s1) + ...
  // This fixes the summation order independently of how the work is launched, so the SAME layer can run as one
  // workgroup per tile (ksplit = 0) or as one workgroup per (tile, segment) writing raw partial tiles to
  // y = partial[segment][M][Cout] (ksplit = 1; splitk_reduce then applies bias / residual / ReLU) with
  // bit-identical results: split-K for small batches without giving up batch invariance.
  // ksplit = 2 ("tail split"): output tiles [0, tail_from) run whole-K as with ksplit = 0 and write y; the tiles from
  // tail_from on (a multiple of ntn: whole rows of tiles, i.e. the rows from (tail_from / ntn) * BM to M) run as
  // (tile, segment) workgroups writing ypart = partial[segment][tail rows][Cout], reduced like ksplit = 1's.  For a batch
  // whose tile count leaves the last round of resident workgroups mostly empty: the remainder is spread over the chip in
  // pieces of one K segment.  Same bits as ksplit = 0 / 1.
  int kseg_len;
  int ksplit;
  int tail_from;
  float *ypart;
  // Walk the output tiles from the last one to the first.  The engine alternates this between consecutive launches:
  // a kernel that starts with the rows its predecessor wrote LAST finds them in the 256-MB Infinity Cache / L2.
  int reverse;
};

enum ConvPrec { kPrecF32 = 0, kPrecBf16x3 = 1, kPrecBf16 = 2 };

enum ConvTile {
  kTileAuto = 0, kTile128x128 = 1, kTile128x64 = 2, kTile64x64 = 3, kTile32x32 = 4,
  kTile128x128w8 = 5,  // 128x128 on 8 waves (512 threads): same LDS as kTile128x128, twice the waves per SIMD
  kTile256x256 = 6,    // conv_bf16_256_kernel: bf16 only, 8 waves, one workgroup per CU, operands by LDS-DMA
  kTileWs = 7,       // conv3x3_ws[128]_kernel: bf16 3x3 s1 p1 with C = Cout = 64 / 128, weights resident in registers, input patch by LDS-DMA
  kTile256x256p = 8, // conv_bf16_256p_kernel: kTile256x256's pipeline run persistently over a workgroup's tiles (K >= 128, Cout <= 2048)
  kNumTiles = 9
};
void conv_tile_dims(int tile, int *bm, int *bn);
// "128x128" | "128x64" | "64x64" | "32x32" | "128x128w8" | "256x256" | "256x256p" | "ws" -> ConvTile (kTileAuto for anything else).
int conv_tile_from_name(const char *name);
// Is `tile` usable for this problem (Cout divisibility)?
bool conv_tile_valid(const ConvParams &p, int tile);
// Do the weight-stationary 3x3 kernels (kTileWs: 64 -> 64 channels, or 128 -> 128) apply to this problem?
bool conv3x3_ws_valid(const ConvParams &p);
bool conv3x3_ws128_valid(const ConvParams &p);
bool conv1x1_wsn_valid(const ConvParams &p);  // 1x1 to 128 / 256 channels (conv1 of layer2 / layer3.0, conv3 + downsample of layer1.0)
bool conv1x1_ws_valid(const ConvParams &p);   // 1x1, 64 / 256 -> 64 channels, optional fused temporal shift (layer1's conv1)

// ks in {1, 3, 7}.  Returns hipSuccess or the launch error.
hipError_t launch_conv(const ConvParams &p, int ks, hipStream_t s);
// Bottleneck.conv2 (3x3, stride 1, pad 1) + bn2 + ReLU + conv3 (1x1) + bn3 + residual + ReLU as ONE launch (fp32 or split-bf16),
// for CMID = 64 / 128 (layer1 / layer2 blocks without a downsample branch).  Bit-identical to launch_conv(conv2)
// followed by launch_conv(conv3 with residual).  prec == kPrecBf16: CMID = 64 only, on the weight-stationary kernel
// (conv3x3_ws_kernel<true>); w3f is then conv3's packed weight matrix [256][64] bf16 itself (no fragment packing).
struct Fused23Params {
  const float *x;      // conv2 input [N, H, W, CMID]
  const float *w2;     // [CMID][9 * CMID]  conv2 weights, K = (ky, kx, c), bn2 scale folded in
  const float *bias2;  // [CMID]
  const float *w3f;    // conv3 weights (bn3 scale folded in) in MFMA-fragment order, tsm_host::pack_w3_fragments[_split]
  const float *bias3;  // [4 * CMID]
  const float *res;    // [M, 4 * CMID]  the block input (identity branch)
  float *y;            // [M, 4 * CMID]
  int N, H, W;
  int M;               // N * H * W
  int kseg_len;        // conv2's K-segment length (ConvParams::kseg_len of that layer; 0 = unsegmented)
  int reverse;         // walk the tiles from the last one to the first (ConvParams::reverse)
};
hipError_t launch_conv23_fused(const Fused23Params &p, int cmid, int prec, hipStream_t s);
// Does the bf16 form apply to n frames of h x w pixels?
bool conv23_ws_valid(int n, int h, int w);

// A whole Bottleneck of layer1 in ONE launch (bf16, bneck_ws_kernel): temporal shift -> conv1 (1x1, cin -> 64) -> conv2 (3x3) ->
// conv3 (1x1, 64 -> 256) + identity -> ReLU.  cin = 256 (layer1.1 / layer1.2): the identity is the block input, w3 = conv3's
// packed weights [256][64]; cin = 64 (layer1.0): the identity is the downsample branch, K-concatenated behind conv3 as in the
// engine's fused conv3 + downsample GEMM: w3 = [256][64 mid | 64 input] and bias3 = conv3's + the downsample's.  Neither
// 64-channel tensor exists in memory and the block input is streamed once.  Bit-identical to the separate launches.
struct BneckParams {
  const void *x;       // [N, H, W, cin] bf16: the block input (conv1's input through the shift, and the identity operand)
  const void *w1;      // [64][cin] bf16
  const float *bias1;  // [64]
  const void *w2;      // [64][576] bf16, K = (ky, kx, c)
  const float *bias2;  // [64]
  const void *w3;      // [256][64] bf16, or [256][128] with the downsample weights behind conv3's (cin = 64)
  const float *bias3;  // [256]
  void *y;             // [N, H, W, 256] bf16
  int cin;             // 256 or 64
  int N, H, W;
  int T, fold;         // temporal shift over T segments (0 = none), fold = cin / shift_div
  int reverse;         // walk the frames from the last one to the first
};
bool bneck_ws_valid(int cin, int n, int h, int w, int T, int fold);
hipError_t launch_bneck_ws(const BneckParams &p, hipStream_t s);

// Temporal shift + conv1 (1x1, 256 -> 128) + bn1 + ReLU + conv2 (3x3, stride 2, pad 1, 128 -> 128) + bn2 + ReLU of layer2.0 as ONE
// launch (bf16, front_s2_kernel, tsm_front.hip): the 128-channel tensor between the two convolutions never exists in memory.
// Bit-identical to launch_conv(conv1 with shift) followed by launch_conv(conv2).
struct FrontParams {
  const void *x;       // [N, H, W, 256] bf16: the block input
  const void *w1;      // [128][256] bf16, bn1 scale folded in
  const float *bias1;  // [128]
  const void *w2;      // [128][1152] bf16, K = (ky, kx, c), bn2 scale folded in
  const float *bias2;  // [128]
  void *y;             // [N, H / 2, (W - 1) / 2 + 1, 128] bf16: conv2's output
  int N, H, W;
  int T, fold;         // temporal shift over T segments (0 = none), fold = 32
  int reverse;         // walk the frames from the last one to the first
};
bool front_s2_valid(int n, int h, int w, int T, int fold);
hipError_t launch_front_s2(const FrontParams &p, hipStream_t s);

// conv3 + bn3 + residual + ReLU of Bottleneck b AND temporal shift + conv1 + bn1 + ReLU of Bottleneck b + 1 as ONE launch
// (bf16, conv31_fused_kernel, tsm_conv31.hip): the block output y is written once (block b + 1's identity) and never read
// back for conv1 -- a tile is all T frames of a clip x 256 / T pixels, so the frames t +- 1 the shifted channels come from
// are rows of the same tile.  Bit-identical to launch_conv(conv3 with residual) followed by launch_conv(conv1 with shift).
struct Conv31Params {
  const void *t2;      // [F * HW, K3] bf16: conv3's input (conv2's output of block b)
  const void *w3;      // [C][K3] bf16, bn3 scale folded in
  const float *bias3;  // [C]
  const void *res;     // [F * HW, C] bf16: block b's input (the identity branch)
  void *y;             // [F * HW, C] bf16: block b's output
  const void *w1;      // [N1][C] bf16: conv1 of block b + 1, bn1 scale folded in
  const float *bias1;  // [N1]
  void *t1;            // [F * HW, N1] bf16: conv1's output of block b + 1
  int n_clips, T, HW;  // F = n_clips * T frames of HW pixels
  int K3, C, N1;
  int fold;            // channels [0, fold) of conv1's input come from frame t + 1, [fold, 2 fold) from t - 1 (0: no shift)
  int reverse;         // walk the tiles from the last one to the first
  int log_px;          // (set by the launcher: log2(256 / T))
};
bool conv31_valid(const Conv31Params &p);
hipError_t launch_conv31_fused(const Conv31Params &p, hipStream_t s);

// Stem (7x7 s2 p3, 3 -> 64) of the bf16 formats as a direct convolution from an LDS-resident pixel-pair patch; x is the
// packed-pair input [n][hi][ceil(wi/2)][8], w the engine's packed stem weights [64][kp], y NHWC, all in `prec`'s format
// (kPrecBf16 or kPrecBf16x3).  Bit-identical to launch_conv(ks = 7) on the same operands.
hipError_t launch_stem_direct(const float *x, const float *w, const float *bias, float *y, int n, int hi, int wi, int kp,
                              int relu, int prec, hipStream_t s);
// The same stem with the 3x3 stride-2 max-pool fused behind it: y is the POOLED tensor [n][hp][wp][64]; the stem's own
// output is never materialised.  Bit-identical to launch_stem_direct followed by launch_maxpool3x3s2.
// planar != 0: x is the [n][3][hi][wi] fp32 tensor (the reference model's input) and the kernel converts it the way
// launch_pack_input would have -- the same bits without the pack launch and its packed copy.
hipError_t launch_stem_pool(const float *x, const float *w, const float *bias, float *y, int n, int hi, int wi, int kp,
                            int relu, int prec, hipStream_t s, int planar = 0);
// Number of K segments of a launch with kseg_len > 0 (1 otherwise).
int conv_num_segments(const ConvParams &p);
// y = act(((p[0] + p[1]) + ...) + bias (+ res)) over fp32 partial tiles [n_seg][M*Cout] written by a ksplit launch.
hipError_t launch_splitk_reduce(const float *partial, int n_seg, int64_t m, int cout, const float *bias,
                                const float *res, float *y, int relu, hipStream_t s);
// Tile rows the heuristics would pick (exposed for tests / DESIGN notes).
void conv_tile_shape(const ConvParams &p, int *bm, int *bn);

// prec == kPrecF32: dst is NHWC4 fp32; bf16 formats: one 8-element group per pixel PAIR (rows of ceil(w/2) groups).
hipError_t launch_pack_input(const float *src, float *dst, int64_t n_frames, int h, int w,
                             int nchw, int prec, hipStream_t s);
// fp32 [n8 * 8 channels] <-> the storage format of `prec` (kPrecBf16x3 or kPrecBf16)
hipError_t launch_from_f32(const float *x, float *y, int64_t n8, int prec, hipStream_t s);
hipError_t launch_to_f32(const float *x, float *y, int64_t n8, int prec, hipStream_t s);
// Fused test transform: [n,h,w,3] u8|f32 frames -> resize (short side -> `resize`, bilinear, no
// antialias, align_corners=False) -> centre crop -> ImageNet normalise -> NHWC4 (out_nchw=0) or NCHW.
struct PreprocParams {
  const void *src;
  float *dst;
  int n, h, w;        // source frames
  int nh, nw;         // resized size
  int top, left, crop;
  int src_is_u8;
  int out_mode;       // 0 = NHWC4 fp32, 1 = NCHW fp32, 2 = NHWC8 split-bf16, 3 = NHWC8 bf16
  float pre_scale;    // 1/255 when frames are to be scaled to [0,1] first, else 1
};
hipError_t launch_preprocess(const PreprocParams &p, hipStream_t s);

// Clips out of a buffer of transformed frames (see gather_clips_kernel).  Buffer frame j holds source frame
// clip_stride * (first_frame + j); pad_frame is the buffer's frame for positions past the end of the video.
struct GatherParams {
  const void *frames;
  void *out;
  int64_t n_frames, frame_bytes, first_frame, total_frames, first_clip, pad_frame;
  int n_clips, n_segment, clip_step, clip_stride;
  int64_t row0;   // first (clip, segment) row of this launch (set by launch_gather_clips: one launch cuts <= 65535 rows)
};
hipError_t launch_gather_clips(const GatherParams &p, hipStream_t s);

hipError_t launch_maxpool3x3s2(const float *x, float *y, int n, int hi, int wi, int c, int prec,
                               hipStream_t s);
hipError_t launch_temporal_shift(const float *x, float *y, int64_t n_frames, int n_segment,
                                 int64_t hw, int c, int fold, hipStream_t s);
// pooled: scratch [n_clips * n_segment, c]
hipError_t launch_head(const float *feat, const float *fc_w, const float *fc_b, float *pooled,
                       float *logits, int n_clips, int n_segment, int hw, int c, int num_class, int prec,
                       hipStream_t s);

// K9: per clip, (softmax,) first arg-max, class id if its score >= threshold else -1; top (nullable) = that score.
hipError_t launch_scores_to_states(const float *logits, int n, int c, int softmax, float threshold, int *states, float *top,
                                   hipStream_t s);

// Launch trace of the calling thread (tsm_trace_launches / tsm_launch_trace in include/tsm_hip.h; tsm_ops.hip).
void note_launch(const char *kernel, const char *where);
void trace_launches(bool on);
const char *launch_trace();   // newline-separated, valid until the thread's next trace call

}  // namespace tsm
