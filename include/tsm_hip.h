/*
 * tsm_hip.h -- C ABI of libtsm_hip.so: TSM-ResNet50 clip inference on MI355X (gfx950).
 *
 * The reference (iucario/WorkoutDetector) has no FFI of its own: the hot path sits behind a
 * Python duck type.  Each entry point below names the reference interface it replaces:
 *
 *   tsm_create / tsm_set_tensor / tsm_finalize
 *       create_model(num_class, num_segments, base_model='resnet50', checkpoint, ...)
 *       workoutdetector/models/tsm.py:422-476  (state-dict keys of TSM, tsm.py:250-262)
 *       and onnxruntime.InferenceSession(ckpt) workoutdetector/utils/inference_count.py:620
 *   tsm_forward
 *       model.run(None, {input_name: float32[1,8,3,224,224]}) -> [float32[1,num_class]]
 *       workoutdetector/utils/inference_count.py:273-275, scripts/eval_classification.py:43-44
 *       == TSM.forward(x[B*T,3,H,W]) -> [B,num_class]   workoutdetector/models/tsm.py:409-419
 *   tsm_tune               (no counterpart: onnxruntime optimises its graph inside InferenceSession(), :620; this is the
 *                          engine's per-batch-size kernel selection, made callable ahead of the first request)
 *   tsm_forward_tap        (parity tests) activation after a named stage of TSM.forward
 *   tsm_temporal_shift     TemporalShift.shift          workoutdetector/models/tsm.py:35-50
 *   tsm_conv_bn_act        one conv + BatchNorm(eval) [+ residual] [+ ReLU] of the torchvision
 *                          Bottleneck kept by TSM        workoutdetector/models/tsm.py:250-251,264-281
 *   tsm_maxpool3x3s2       base_model.maxpool
 *   tsm_head               avgpool -> fc -> view(-1,T,cls) -> mean(1)   tsm.py:411-419,165-174
 *   tsm_gather_clips       the loop's clip windows: video[i:i + 16:2] for i in range(0, len(video), 8), zero-padded tail
 *   tsm_scores_to_states   per clip: to_softmax, first arg-max, score >= 0.5 ? class : -1
 *                          workoutdetector/utils/eval.py:153-164, utils/visualize.py:140-150
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / HIP types in signatures.  hip streams travel as
 *     void*.  NULL means: the device's default (null) stream for TSM_MEM_DEVICE calls and per-op
 *     entry points (so work is ordered after whatever produced the buffers there -- torch's default
 *     stream is the null stream), the engine's private stream for TSM_MEM_HOST calls.
 *   - Every function returns 0 on success or a negative tsm_status; the message for the last
 *     failure on an engine is tsm_last_error(engine) (engine == NULL: the calling THREAD's last failure of
 *     tsm_create or of an engine-less per-op entry point; no process-global state).
 *   - An engine owns its weights and workspace on ONE device; it is NOT re-entrant: one
 *     in-flight call per engine.  Independent engines (other devices / processes) coexist.
 *   - Caller owns all input / output buffers.  With TSM_MEM_HOST the call copies and
 *     synchronises before returning; with TSM_MEM_DEVICE the call only enqueues on `stream` -- with ONE
 *     exception: the FIRST tsm_forward of a new power-of-two bucket of n_clips tunes its kernels first (unless
 *     TSM_AUTOTUNE=0, or TSM_TUNE_CACHE names a file that already holds this bucket): it times every candidate on the
 *     real launches, which synchronises with `stream` about a hundred times and takes a few hundred ms, so that call is
 *     neither asynchronous nor legal inside a stream capture.  Call tsm_tune per bucket at start-up (TsmEngine.warmup
 *     in the Python host) before capturing a graph or relying on enqueue-only behaviour; every later call of that
 *     bucket allocates nothing, synchronises nothing and is capture-safe.
 *   - Activations inside the engine are NHWC fp32.
 */
#ifndef TSM_HIP_H_
#define TSM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSM_ABI_VERSION 7 /* 7: tsm_build_id, tsm_trace_launches / tsm_launch_trace; 6: tsm_tune, per-user default tune cache; 5: tsm_gather_clips; 4: tsm_scores_to_states; tile codes lost the tail field; TSM_* variables read in tsm_create only */

typedef enum tsm_status {
  TSM_OK = 0,
  TSM_ERR_INVALID_ARG = -1,
  TSM_ERR_HIP = -2,
  TSM_ERR_NOT_FINALIZED = -3,
  TSM_ERR_MISSING_TENSOR = -4,
  TSM_ERR_SHAPE = -5,
  TSM_ERR_CAPACITY = -6,
  TSM_ERR_UNSUPPORTED = -7
} tsm_status;

typedef enum tsm_memkind { TSM_MEM_HOST = 0, TSM_MEM_DEVICE = 1 } tsm_memkind;

/* Layout of the clip tensor handed to tsm_forward (T = num_segments). */
typedef enum tsm_layout {
  TSM_LAYOUT_NTCHW = 0, /* float32 [B,T,3,H,W]  -- the reference's ONNX input; the stem kernel reads it as it is
                           (rounds / splits while it stages its patch: no packed copy, no extra launch)           */
  TSM_LAYOUT_NTHWC = 1, /* float32 [B,T,H,W,3]  -- decoder-native, skips the host permute */
  TSM_LAYOUT_NTHWC4 = 2, /* float32 [B,T,H,W,4]  -- what tsm_preprocess writes for a TSM_DTYPE_F32 engine
                            (4th channel 0: the stem never reads it); device memory only, consumed in place without a repack */
  TSM_LAYOUT_NTHWC8S = 3, /* split-bf16 [B,T,H,ceil(W/2),8]: one 32-byte group [hi x8 | lo x8] per PIXEL PAIR,
                             elements (pixel 2j: c0 c1 c2 0, pixel 2j+1: c0 c1 c2 0), an odd width ends in a
                             zero pixel -- what tsm_preprocess writes for a TSM_DTYPE_BF16X3 engine (the 7x7
                             stride-2 stem then reads 4 aligned pairs per kernel row); device memory only */
  TSM_LAYOUT_NTHWC8B = 4  /* bf16 [B,T,H,ceil(W/2),8]: the same pixel pairs, 16 bytes per pair -- for a
                             TSM_DTYPE_BF16 engine */
} tsm_layout;

typedef enum tsm_pixel { TSM_PIXEL_U8 = 0, TSM_PIXEL_F32 = 1 } tsm_pixel;

/* Arithmetic of the conv stack.
 *   TSM_DTYPE_F32     fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): bit-for-bit an fp32 fma chain.
 *   TSM_DTYPE_BF16X3  "split-bf16": every activation/weight is kept as hi = bf16(x), lo = bf16(x - hi)
 *                     (4 bytes per element, like fp32) and a*b is computed as ah*bh + ah*bl + al*bh on
 *                     the bf16 MFMA with fp32 accumulation: ~2^-17 relative error per product (fp32-class
 *                     results, within the path's rtol 1e-3 with >10x margin) at several times the speed.
 *   TSM_DTYPE_BF16    bf16 weights and activations, one bf16 MFMA per product, fp32 accumulate
 *                     (BASELINE.json config 5, the roofline stress config).  NOT within rtol 1e-3 of the
 *                     fp32 reference: ~1e-2 relative on logits; tests state the tolerance. */
typedef enum tsm_dtype { TSM_DTYPE_F32 = 0, TSM_DTYPE_BF16X3 = 1, TSM_DTYPE_BF16 = 2 } tsm_dtype;

typedef struct tsm_config {
  int32_t struct_size;  /* = sizeof(tsm_config), ABI guard                           */
  int32_t num_class;    /* 12 for the RepCount 6-action x 2-state model              */
  int32_t num_segments; /* T, 8 (16 for the stress config)                           */
  int32_t height;       /* 224                                                       */
  int32_t width;        /* 224                                                       */
  int32_t shift_div;    /* 8: fold = C / shift_div                                   */
  int32_t is_shift;     /* 1: temporal shift in front of every Bottleneck.conv1      */
  int32_t max_clips;    /* workspace capacity in clips per tsm_forward call          */
  int32_t device_id;    /* HIP device ordinal                                        */
  int32_t dtype;        /* TSM_DTYPE_F32 | TSM_DTYPE_BF16X3 | TSM_DTYPE_BF16         */
} tsm_config;

typedef struct tsm_engine tsm_engine;

int tsm_abi_version(void);

/* The identity of the SOURCE this binary was built from: the first 16 hex digits of the sha256 over csrc/ (file names,
 * contents, per-file compiler options) and the extra compiler definitions of the build -- what
 * workoutdetector_amd/build.py::build_id() computes from the tree.  The library is git-ignored and travels prebuilt, so
 * the Python host refuses one whose id is not the tree's (workoutdetector_amd/_lib.py) and bench.py prints it; it also
 * keys the tune cache.  A hand build without -DTSM_BUILD_ID reports its compile time stamp.
 * (No counterpart in the reference: a Python package cannot be stale against itself.) */
const char *tsm_build_id(void);

/* Launch trace of the CALLING THREAD (parity tests: "did the kernel under test really run?").  tsm_trace_launches(1)
 * clears the thread's trace and starts recording one line per kernel launch this library makes from that thread -- the
 * kernel's name as rocprofv3 would print it up to template-argument spelling, e.g. "bneck_ws_kernel<256, true, true>" or
 * "conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, kPrecBf16> [BM = 64, BN = 64, ...]"; tsm_trace_launches(0) stops.
 * tsm_launch_trace copies the newline-separated trace (NUL-terminated) into buf when cap suffices and always returns the
 * bytes needed.  Off by default; per thread, like the engine-less error message: no process-global state.
 * (No counterpart in the reference: onnxruntime's session.run is opaque, utils/inference_count.py:273-275.) */
int tsm_trace_launches(int32_t on);
int64_t tsm_launch_trace(char *buf, int64_t cap);

/* Engine lifetime ------------------------------------------------------------------------- */
int tsm_create(const tsm_config *cfg, tsm_engine **out);
void tsm_destroy(tsm_engine *e);
const char *tsm_last_error(const tsm_engine *e);

/* Hand one state-dict tensor to the engine (host memory, float32, torch layout: conv OIHW,
 * BN vectors [C], fc [num_class, 2048]).  Names are the reference's TSM.state_dict() keys, e.g.
 * "base_model.layer1.0.conv1.net.weight" ("...conv1.weight" is accepted too).  The engine copies;
 * the caller keeps ownership.  Unknown names return TSM_ERR_INVALID_ARG. */
int tsm_set_tensor(tsm_engine *e, const char *name, const float *host_data, const int64_t *shape,
                   int32_t ndim);
/* Fold BatchNorm into the convs, pack to K-major NHWC tiles, upload, allocate the workspace. */
int tsm_finalize(tsm_engine *e);

/* Hot path ---------------------------------------------------------------------------------
 * clips:  n_clips x T x 3 x H x W float32 in `layout`, in `memkind` memory.
 * logits: float32 [n_clips, num_class] in the same memkind.  Raw scores (before softmax). */
int tsm_forward(tsm_engine *e, const void *clips, int32_t memkind, int32_t layout, int32_t n_clips,
                float *logits, void *stream);

/* Tune the kernels of the bucket `n_clips` falls into NOW (synchronous: a few hundred ms of timed launches on the
 * engine's own zeroed input buffer; no caller memory is touched), or read the choices from the tune cache file: what
 * the first tsm_forward of that bucket would otherwise do inside the call.  A no-op when the bucket is already tuned or
 * TSM_AUTOTUNE=0.  Call it at start-up -- a service before it takes requests, a dataset job while its first frames are
 * still being decoded (workoutdetector_amd/inference_count.py does) -- for every batch size that will occur; every
 * tsm_forward(TSM_MEM_DEVICE) afterwards only enqueues.
 * Tune cache: TSM_TUNE_CACHE=<file>, default $XDG_CACHE_HOME/tsm_hip/tune_cache.txt (else ~/.cache/tsm_hip/...); lines
 * are keyed by ABI, library build, device, geometry and dtype, so a second process (or the other ranks of a job) skips
 * the timing pass; TSM_TUNE_CACHE= (empty), 0 or off disables the file. */
int tsm_tune(tsm_engine *e, int32_t n_clips, void *stream);

/* Same as tsm_forward but stops after `stage` and returns that activation (NHWC fp32) in
 * `out` (capacity in floats); shape [N*T, H, W, C] written to out_shape[4].
 * Stages: "input" (packed NHWC4), "conv1" (stem conv+bn+relu), "stem" (after maxpool),
 * "layer{1..4}.{b}" (block output), "layer{L}.{b}.conv1|conv2" (branch intermediates). */
int tsm_forward_tap(tsm_engine *e, const void *clips, int32_t memkind, int32_t layout,
                    int32_t n_clips, const char *stage, float *out, int64_t out_capacity,
                    int64_t out_shape[4], void *stream);

/* Kernel time of the most recent tsm_forward on this engine, measured with HIP events on the
 * stream the kernels ran on (ms); negative if none.  Synchronises on the stop event. */
float tsm_last_forward_ms(tsm_engine *e);

/* Per-launch timing for bench.py's roofline: keep HIP-event pairs around the kernel launches of the
 * next `n_forwards` tsm_forward calls (0 switches it off; at most 64 are kept).  `only_conv3x3` != 0
 * limits the pairs to the 3x3 convolutions (the dominant kernel), which keeps the marker overhead
 * inside a timed region below 0.5 %; launches without a pair report -1.  tsm_layer_times
 * synchronises on forward `forward_index` (0-based since the last tsm_set_layer_timing) and writes
 * one duration in ms per launch, in launch order: pack_input, conv1 (stem), maxpool, then per block
 * [downsample,] conv1, conv2, conv3, then head (pool + fc).  *n_out = number of launches. */
int tsm_set_layer_timing(tsm_engine *e, int32_t n_forwards, int32_t only_conv3x3);
int tsm_layer_times(tsm_engine *e, int32_t forward_index, float *ms_out, int32_t cap, int32_t *n_out);

/* Conv tile code the engine's autotuner chose for each conv launch of an `n_clips` forward, in launch order (stem,
 * then per block [downsample,] conv1, conv2, conv3): 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 32x32 (one wave),
 * 5 = 128x128 on 8 waves, 6 = 256x256 LDS-DMA kernel (bf16), 7 = weight-stationary 3x3 (bf16, 64 -> 64 / 128 -> 128
 * channels), 8 = the 256x256 kernel run persistently over a workgroup's tiles (bf16, K >= 128), 0 = not tuned (heuristic); + 256 = split-K form of a segmented fp32 layer (one workgroup per tile and K
 * segment, combined in segment order); + 1024 (on conv2's code) = the block runs conv2 + conv3 + residual as ONE launch
 * (conv3's slot is then not used); + 2048 (on conv1's code) = the WHOLE block -- shift, conv1, conv2, conv3 (+ the fused
 * downsample branch) + identity -- runs as ONE launch (bf16 layer1; the conv2 / conv3 slots are then not used); + 4096 (on
 * conv3's code) = that launch also runs the temporal shift + conv1 of the NEXT block (bf16 layer2; the next block's conv1
 * slot is then not used).
 * The first tsm_forward with a new power-of-two bucket of n_clips
 * times every valid code per layer once, SYNCHRONOUSLY (see Conventions: not capture-safe, a few hundred ms; results are
 * bit-identical across codes); TSM_AUTOTUNE=0 in the environment at tsm_create disables it, TSM_TUNE_CACHE=<file> lets a
 * later process (or the other ranks of a job) read the choices instead of timing again.  Every TSM_* environment
 * variable is read once, in tsm_create. */
int tsm_conv_tiles(tsm_engine *e, int32_t n_clips, int32_t *tiles_out, int32_t cap, int32_t *n_out);

/* Per-op entry points (device pointers; used by the parity tests and as building blocks) ----- */

/* NHWC temporal shift, x/y: [n_frames, hw, c]; n_frames % n_segment == 0; c % (4*fold_div)==0 */
int tsm_temporal_shift(const float *x, float *y, int64_t n_frames, int32_t n_segment, int64_t hw,
                       int32_t c, int32_t fold_div, void *stream);

/* y = act( conv(x, w) * bn_scale + bn_bias [+ residual] ), NHWC.
 * x [n,hi,wi,cin]; w OIHW [cout,cin,k,k] (device, raw); gamma/beta/mean/var [cout] (device);
 * k in {1,3,7}; pad = k/2; residual (nullable) and y [n,ho,wo,cout].
 * shift_segments > 0 applies the temporal shift (fold_div) to x on the fly (k == 1, stride 1).
 * dtype: any tsm_dtype (x / residual / y stay fp32 NHWC at this boundary and are converted to and from
 * the storage format of that dtype around the kernel).
 * Packs the weights on every call: a test/debug entry point, not the fast path.  It has no engine, so it is the one
 * place that reads tuning variables from the environment PER CALL: TSM_CONV_TILE (force a tile shape where valid) and
 * TSM_STEM_DIRECT (bf16-format stems on the generic kernel); neither changes a result bit. */
int tsm_conv_bn_act(const float *x, const float *w, const float *gamma, const float *beta,
                    const float *mean, const float *var, const float *residual, float *y,
                    int32_t n, int32_t hi, int32_t wi, int32_t cin, int32_t cout, int32_t k,
                    int32_t stride, int32_t relu, int32_t shift_segments, int32_t fold_div,
                    int32_t dtype, void *stream);

int tsm_maxpool3x3s2(const float *x, float *y, int32_t n, int32_t hi, int32_t wi, int32_t c,
                     void *stream);

/* Fused test transform on the GPU (device pointers), the pre-step of the hot path:
 *   build_test_transform(person_crop=False) = ConvertImageDtype -> Resize(resize) -> CenterCrop(crop)
 *   -> Normalize(ImageNet)            workoutdetector/datasets/build.py:131-136
 * frames: [n, h, w, 3] decoder layout, TSM_PIXEL_U8 or TSM_PIXEL_F32 (values 0..255).
 * out:    out_layout TSM_LAYOUT_NTHWC4 -> [n, crop, crop, 4] fp32, TSM_LAYOUT_NTHWC8S -> split-bf16 /
 *         TSM_LAYOUT_NTHWC8B -> bf16 pixel pairs [n, crop, ceil(crop/2), 8] (feed tsm_forward of an engine
 *         of the matching dtype directly) or TSM_LAYOUT_NTCHW -> [n, 3, crop, crop] fp32.
 * scale_255 = 0 reproduces the reference's inference_dataset, which never divides by 255
 * (utils/inference_count.py:412-414); 1 scales to [0,1] first. */
int tsm_preprocess(const void *frames, int32_t pixel, int32_t n, int32_t h, int32_t w, float *out,
                   int32_t out_layout, int32_t resize, int32_t crop, int32_t scale_255, void *stream);

/* The clip iterator of the dataset loop on the GPU (device pointers), between tsm_preprocess and tsm_forward:
 *   for i in range(0, len(video), 8): clip = video[i:i + 16:2], the tail zero-padded   utils/inference_count.py:411-414
 * over TRANSFORMED frames: the buffer `frames` [n_frames, frame_bytes] (any tsm_preprocess layout; rows are opaque) holds
 * every clip_stride-th frame of the video from source frame clip_stride * first_frame on, i.e. buffer frame j = source
 * frame clip_stride * (first_frame + j).  out [n_clips, n_segment, frame_bytes]:
 *   out[c][k] = source frame clip_step * (first_clip + c) + clip_stride * k   if that index < total_frames,
 *               buffer frame pad_frame                                         otherwise (the transformed zero frame).
 * Every index is validated on the host before the launch (TSM_ERR_INVALID_ARG, nothing launched): frame_bytes % 16 == 0,
 * clip_step % clip_stride == 0, each clip starts inside the video, all frames it reads lie in the buffer, and -- when the
 * range has a padded tail -- pad_frame lies in the buffer and is NOT one of the frames the range reads as video frames.
 * n_clips is not limited by the launch geometry: ranges of more than 65535 (clip, segment) rows are cut into several
 * launches inside the call. */
int tsm_gather_clips(const void *frames, int64_t n_frames, int64_t frame_bytes, int64_t first_frame, int64_t total_frames,
                     int64_t pad_frame, int64_t first_clip, int32_t n_clips, int32_t n_segment, int32_t clip_step,
                     int32_t clip_stride, void *out, void *stream);

/* feat [n_clips*T, hw, c] NHWC -> logits [n_clips, num_class]; fc_w [num_class, c], fc_b. */
int tsm_head(const float *feat, const float *fc_w, const float *fc_b, float *logits,
             int32_t n_clips, int32_t n_segment, int32_t hw, int32_t c, int32_t num_class,
             void *stream);

/* Scores -> states on the GPU (device pointers), the post-step of the hot path:
 *   logits [n_clips, num_class] fp32 -> states [n_clips] int32: (softmax != 0: fp32 softmax over the classes,) the FIRST
 *   maximum, its class id if the score >= threshold, else -1; top_score (nullable) [n_clips] receives that score.
 * A streaming consumer (count_by_video_model, utils/inference_count.py:285-339) then copies 4-8 bytes per window to
 * the host instead of the logits and feeds pred_to_count directly. */
int tsm_scores_to_states(const float *logits, int32_t n_clips, int32_t num_class, int32_t softmax, float threshold,
                         int32_t *states, float *top_score, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSM_HIP_H_ */
