// Host engine + C ABI (include/tsm_hip.h) for the TSM-ResNet50 clip forward on MI355X.
//
// Owns: packed weights (BatchNorm folded, K-major), an NHWC fp32 activation workspace sized for
// max_clips, one HIP stream, two timing events.  The forward is a fixed schedule of kernel launches
// (csrc/tsm_*.hip, one file per kernel family: tsm_device.h lists them); nothing here falls back to a CPU path.
//
// Reference behaviour mirrored: workoutdetector/models/tsm.py:409-419 (TSM.forward), :125-137
// (shift in front of every Bottleneck.conv1), :451-473 (state-dict naming); torchvision-0.13
// ResNet-50 v1.5 (stride on conv2, BN eps 1e-5).
#include <hip/hip_runtime.h>

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tsm_hip.h"
#include "tsm_host_util.h"
#include "tsm_kernels.h"

#ifndef TSM_BUILD_ID   // workoutdetector_amd/build.py passes the sha of csrc/; a hand build falls back to its own time stamp
#define TSM_BUILD_ID __DATE__ "/" __TIME__
#endif

namespace {

using namespace tsm_host;

// The tag is also what workoutdetector_amd/build.py looks for in the FILE to decide whether a prebuilt library belongs to
// the tree it sits in (mtimes do not survive a copy to another machine).
const char kBuildTag[] = "tsm-build-id:" TSM_BUILD_ID;

constexpr int kBlocks[4] = {3, 4, 6, 3};
constexpr int kPlanes[4] = {64, 128, 256, 512};

// Message of the last failure of an engine-less entry point (tsm_create, the per-op functions): per calling thread, so
// two threads driving two engines never write the same string (include/tsm_hip.h, "no global state").
thread_local std::string g_create_error;

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct ConvLayer {
  std::string wkey, bnp;
  int cin = 0, cout = 0, k = 1, stride = 1;
  int cp = 0;   // channel count the kernel sees (stem: 3 -> 4)
  int kp = 0;   // padded K
  int kseg = 0; // K-steps per accumulation segment (fp32 layers with long K, ConvParams::kseg_len); 0 = unsegmented
  float *d_w = nullptr, *d_b = nullptr;
};

struct Block {
  int conv1, conv2, conv3, down;  // indices into convs, down = -1 if none
  int stride;
  // conv3 + downsample as ONE GEMM over K = [conv3 input channels | block input channels]
  float *d_wf = nullptr, *d_bf = nullptr;
  int kpf = 0;
  int ksegf = 0;  // segment length of the fused GEMM
  // conv2 + conv3 (+ residual) as ONE kernel (tsm::launch_conv23_fused): fp32 / split-bf16 blocks without a downsample branch whose
  // mid tensor has 64 / 128 channels (layer1.1-2, layer2.1-3); d_w3f = conv3's folded weights in fragment order
  // (fp32 / split-bf16; the bf16 form, 64 channels only, reads conv3's own packed matrix: d_w3f stays null, cmid = 64)
  float *d_w3f = nullptr;
  int cmid = 0;
};

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

struct tsm_engine {
  tsm_config cfg{};
  std::string err;
  hipStream_t stream = nullptr;
  bool finalized = false;
  int prec = tsm::kPrecF32;
  float *d_tap = nullptr;     // fp32 staging for tsm_forward_tap on split-format engines
  std::map<std::string, HostTensor> tensors;
  std::vector<ConvLayer> convs;
  std::vector<Block> blocks;
  float *d_fcw = nullptr, *d_fcb = nullptr;
  float *d_in = nullptr;      // raw clips staged from host memory
  float *d_in4 = nullptr;     // NHWC4 packed input
  float *buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float *d_pooled = nullptr, *d_logits = nullptr;
  float *d_partial = nullptr;   // split-K segment sums [segments][M][Cout] (fp32 engines)
  size_t partial_elems = 0;
  size_t buf_elems = 0;
  int h1 = 0, w1 = 0, hp = 0, wp = 0;  // stem conv / maxpool output sizes
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_time = false;
  std::vector<void *> allocs;
  // per-launch timing (tsm_set_layer_timing): timing[f] = events of forward f, 2 per launch
  // conv tile autotune: per frame count, one ConvTile per conv layer (0 = not tuned yet)
  std::map<int, std::vector<int>> tile_cache;
  bool autotune = true;
  bool fuse_down = true;  // TSM_FUSE_DOWNSAMPLE=0 runs the downsample branch as its own launch
  bool stem_pool = true;    // TSM_STEM_POOL=0: separate max-pool launch behind the direct stem
  bool stem_direct = true;  // TSM_STEM_DIRECT=0: bf16-format stems on the generic implicit-GEMM kernel (bit-identical, slower)
  bool stem_planar = true;  // TSM_STEM_PLANAR=0: an NTCHW input is packed by its own launch first (bit-identical, one more pass over the input)
  // TSM_TUNE_CACHE=<file>: tuned tile codes are appended to / read from this file, one line per bucket, keyed by
  // `tune_sig` (ABI, device name, geometry, dtype): a later process skips the timing pass.  Codes never change
  // results and every code is re-validated against its layer at launch (run_forward, code_ok), so a stale, foreign
  // or hand-edited line can only cost speed; malformed lines are ignored.
  std::string tune_path, tune_sig;
  // Tuning hooks, read ONCE in tsm_create (never per launch): TSM_CONV_TILE=<name> forces one tile shape wherever it
  // is valid, TSM_CONV_CODE=<int> one tile code (tile | 0x100 = split-K form); tests and tools/ sweeps only.
  int force_tile = 0, force_code = -1;
  int tail_split = 1;   // TSM_TAIL_SPLIT=0: the tuner does not try the tail-split form (tile code bit 0x200)
  // Consecutive conv launches walk their output tiles in opposite directions: a kernel starts with the rows its
  // predecessor wrote last, which are still in the Infinity Cache / L2 (bit-neutral; -1...2 % forward time in the
  // HBM-bound formats, nothing in fp32).  TSM_ZIGZAG=0 switches it off.
  bool zigzag = true;
  int fuse23 = -1;   // TSM_FUSE_CONV23: 0 never, 1 wherever a block is eligible, unset: the autotuner times both forms
  int fuse_block = -1;   // TSM_FUSE_BLOCK: the same for the whole-Bottleneck kernel (bf16, layer1.1 / layer1.2)
  int fuse31 = -1;       // TSM_FUSE_C3C1: the same for conv3 of block b + shift + conv1 of block b + 1 as one launch (bf16, layer2)
  int fuse_front = -1;   // TSM_FUSE_FRONT: the same for shift + conv1 + the stride-2 conv2 of layer2.0 as one launch (bf16)
  int n_cu = 256;
  int timing_left = 0;
  bool timing_only3x3 = false;
  std::vector<std::vector<hipEvent_t>> timing;
  std::vector<hipEvent_t> event_pool;
  std::vector<hipEvent_t> *cur_timing = nullptr;
};

namespace {

int fail(tsm_engine *e, int code, const std::string &msg) {
  if (e) e->err = msg; else g_create_error = msg;
  return code;
}

#define TSM_HIP(e, call)                                                                        \
  do {                                                                                          \
    hipError_t _st = (call);                                                                    \
    if (_st != hipSuccess)                                                                      \
      return fail((e), TSM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st));        \
  } while (0)

void build_topology(tsm_engine *e) {
  e->convs.clear();
  e->blocks.clear();
  ConvLayer stem;
  stem.wkey = "base_model.conv1.weight";
  stem.bnp = "base_model.bn1";
  stem.cin = 3; stem.cout = 64; stem.k = 7; stem.stride = 2;
  stem.cp = 4;
  // fp32: K = 49 taps x 4 channels; bf16 formats: K = 7 rows x 4 pixel pairs x 8 (fold_and_pack_stem_pairs)
  stem.kp = e->prec == tsm::kPrecF32 ? round_up(7 * 7 * 4, 32) : round_up(7 * 4 * 8, e->prec == tsm::kPrecBf16 ? 64 : 32);
  e->convs.push_back(stem);
  int cin = 64;
  for (int li = 0; li < 4; ++li) {
    for (int b = 0; b < kBlocks[li]; ++b) {
      const int planes = kPlanes[li];
      const int stride = (b == 0 && li > 0) ? 2 : 1;
      const std::string p = "base_model.layer" + std::to_string(li + 1) + "." + std::to_string(b);
      Block blk;
      blk.stride = stride;
      ConvLayer c1; c1.wkey = p + ".conv1.net.weight"; c1.bnp = p + ".bn1";
      c1.cin = cin; c1.cout = planes; c1.k = 1; c1.stride = 1; c1.cp = cin; c1.kp = cin;
      ConvLayer c2; c2.wkey = p + ".conv2.weight"; c2.bnp = p + ".bn2";
      c2.cin = planes; c2.cout = planes; c2.k = 3; c2.stride = stride; c2.cp = planes; c2.kp = 9 * planes;
      ConvLayer c3; c3.wkey = p + ".conv3.weight"; c3.bnp = p + ".bn3";
      c3.cin = planes; c3.cout = planes * 4; c3.k = 1; c3.stride = 1; c3.cp = planes; c3.kp = planes;
      c1.kseg = segment_len(c1.kp, e->prec);
      c2.kseg = segment_len(c2.kp, e->prec);
      blk.conv1 = (int)e->convs.size(); e->convs.push_back(c1);
      blk.conv2 = (int)e->convs.size(); e->convs.push_back(c2);
      blk.conv3 = (int)e->convs.size(); e->convs.push_back(c3);
      blk.down = -1;
      if (b == 0) {
        ConvLayer d; d.wkey = p + ".downsample.0.weight"; d.bnp = p + ".downsample.1";
        d.cin = cin; d.cout = planes * 4; d.k = 1; d.stride = stride; d.cp = cin; d.kp = cin;
        blk.down = (int)e->convs.size(); e->convs.push_back(d);
      }
      e->blocks.push_back(blk);
      cin = planes * 4;
    }
  }
}

const HostTensor *find_tensor(const tsm_engine *e, const std::string &key) {
  auto it = e->tensors.find(key);
  if (it != e->tensors.end()) return &it->second;
  // TemporalShift wraps conv1 as ".conv1.net.weight"; accept the un-wrapped spelling too.
  const std::string net = ".conv1.net.weight";
  if (key.size() > net.size() && key.compare(key.size() - net.size(), net.size(), net) == 0) {
    std::string alt = key.substr(0, key.size() - net.size()) + ".conv1.weight";
    it = e->tensors.find(alt);
    if (it != e->tensors.end()) return &it->second;
  }
  return nullptr;
}

bool known_name(const tsm_engine *e, const std::string &name) {
  if (name == "fc.weight" || name == "fc.bias") return true;
  static const char *bn_suffix[] = {".weight", ".bias", ".running_mean", ".running_var", ".num_batches_tracked"};
  for (const ConvLayer &c : e->convs) {
    if (name == c.wkey) return true;
    const std::string net = ".conv1.net.weight";
    if (c.wkey.size() > net.size() && c.wkey.compare(c.wkey.size() - net.size(), net.size(), net) == 0 &&
        name == c.wkey.substr(0, c.wkey.size() - net.size()) + ".conv1.weight")
      return true;
    for (const char *s : bn_suffix)
      if (name == c.bnp + s) return true;
  }
  return false;
}

int dev_alloc(tsm_engine *e, float **p, size_t elems) {
  void *q = nullptr;
  TSM_HIP(e, hipMalloc(&q, elems * sizeof(float)));
  e->allocs.push_back(q);
  *p = static_cast<float *>(q);
  return TSM_OK;
}

tsm::ConvParams make_params(const ConvLayer &c, const float *x, const float *res, float *y, int n, int hi,
                            int wi, bool relu, int T, int shift_div, int prec = tsm::kPrecF32) {
  tsm::ConvParams p{};
  p.prec = prec;
  p.x = x; p.w = c.d_w; p.bias = c.d_b; p.res = res; p.y = y;
  p.N = n; p.Hi = hi; p.Wi = wi; p.C = c.cp; p.logC4 = ilog2(c.cp / 4);
  p.pad = c.k / 2; p.stride = c.stride;
  p.Ho = (hi + 2 * p.pad - c.k) / c.stride + 1;
  p.Wo = (wi + 2 * p.pad - c.k) / c.stride + 1;
  p.Cout = c.cout; p.Kp = c.kp; p.M = n * p.Ho * p.Wo; p.relu = relu ? 1 : 0;
  p.T = T; p.fold = T > 0 ? c.cp / shift_div : 0;
  p.kseg_len = res ? 0 : c.kseg;   // (no layer with a residual has a long K; the per-op entry point passes kseg = 0)
  return p;
}

// $XDG_CACHE_HOME/tsm_hip/tune_cache.txt, else $HOME/.cache/tsm_hip/tune_cache.txt ("" when neither is set or the
// directory cannot be made: no cache, never an error).
std::string default_tune_cache_path() {
  std::string base;
  if (const char *x = getenv("XDG_CACHE_HOME")) base = x;
  if (base.empty()) {
    const char *h = getenv("HOME");
    if (!h || !*h) return "";
    base = std::string(h) + "/.cache";
    (void)mkdir(base.c_str(), 0700);
  }
  base += "/tsm_hip";
  if (mkdir(base.c_str(), 0700) != 0 && errno != EEXIST) return "";
  return base + "/tune_cache.txt";
}

bool tune_cache_load(tsm_engine *e, int key, std::vector<int> *codes) {
  if (e->tune_path.empty()) return false;
  FILE *f = fopen(e->tune_path.c_str(), "r");
  if (!f) return false;
  const std::string want = e->tune_sig + "|" + std::to_string(key) + "|";
  char line[4096];
  bool ok = false;
  while (!ok && fgets(line, sizeof line, f)) ok = parse_tune_line(line, want, tsm::kNumTiles, codes);
  fclose(f);
  return ok;
}

void tune_cache_store(tsm_engine *e, int key, const std::vector<int> &codes) {
  if (e->tune_path.empty()) return;
  std::string line = e->tune_sig + "|" + std::to_string(key) + "|";
  for (size_t i = 0; i < codes.size(); ++i) line += (i ? "," : "") + std::to_string(codes[i]);
  line += "\n";
  if (line.size() >= 4096) return;
  FILE *f = fopen(e->tune_path.c_str(), "a");   // one short write per line: appends from several ranks do not interleave
  if (!f) return;
  fwrite(line.data(), 1, line.size(), f);
  fclose(f);
}

// The tail split of a segmented 64x64 launch (tile code bit 0x200): tsm_host_util.h::tail_split_point on this launch's shape.
bool tail_split_from(const tsm::ConvParams &p, int n_cu, size_t partial_elems, int *tail_from) {
  if (p.kseg_len <= 0) return false;
  const long from = tsm_host::tail_split_point(p.M, p.Cout, tsm::conv_num_segments(p), n_cu, partial_elems);
  *tail_from = (int)from;
  return from > 0;
}

struct Tap {
  const float *ptr = nullptr;
  int64_t shape[4] = {0, 0, 0, 0};
  bool hit = false;
};

// One event before and one after a launch when per-launch timing is armed for this forward.
struct LaunchTimer {
  tsm_engine *e;
  hipStream_t s;
  bool on;
  bool skip;  // armed, but this launch is outside the selection: keep the slot, record nothing
  LaunchTimer(tsm_engine *e_, hipStream_t s_, bool is3x3)
      : e(e_), s(s_), on(e_->cur_timing != nullptr), skip(e_->timing_only3x3 && !is3x3) { mark(); }
  ~LaunchTimer() { mark(); }
  void mark() {
    if (!on) return;
    if (skip) {
      e->cur_timing->push_back(nullptr);
      return;
    }
    hipEvent_t ev = nullptr;
    if (!e->event_pool.empty()) {
      ev = e->event_pool.back();
      e->event_pool.pop_back();
    } else if (hipEventCreate(&ev) != hipSuccess) {
      return;
    }
    (void)hipEventRecord(ev, s);
    e->cur_timing->push_back(ev);
  }
};
#define TSM_LAUNCH_K(e, s, is3x3, call) \
  do {                                  \
    LaunchTimer _lt((e), (s), (is3x3)); \
    TSM_HIP((e), (call));               \
  } while (0)
#define TSM_LAUNCH(e, s, call) TSM_LAUNCH_K(e, s, false, call)

// Enqueue the forward on `s`.  If `stage` is non-null, stop right after that stage and report it.
int run_forward(tsm_engine *e, const float *d_clips, int layout, int n_clips, float *d_logits,
                hipStream_t s, const char *stage, Tap *tap) {
  const tsm_config &cfg = e->cfg;
  const int T = cfg.num_segments, n = n_clips * T;
  const int shiftT = cfg.is_shift ? T : 0;
  auto want = [&](const std::string &name) { return stage && name == stage; };
  auto hit = [&](const float *p, int64_t a, int64_t b, int64_t c, int64_t d) {
    tap->ptr = p; tap->shape[0] = a; tap->shape[1] = b; tap->shape[2] = c; tap->shape[3] = d; tap->hit = true;
    return TSM_OK;
  };

  // Tile shape per conv layer for this frame count: tuned on first use by timing every valid shape on
  // the real launch (all shapes give bit-identical results: the k order per output does not depend on
  // the tile), then cached.  TSM_AUTOTUNE=0 keeps the static heuristic.
  std::vector<int> *tiles = nullptr;
  bool tuning = false;
  if (e->autotune && !stage) {
    auto it = e->tile_cache.find(tile_bucket(n_clips) * T);
    if (it == e->tile_cache.end()) {
      it = e->tile_cache.emplace(tile_bucket(n_clips) * T, std::vector<int>(e->convs.size(), 0)).first;
      tuning = true;
    }
    tiles = &it->second;
  }
  // A tile code is a ConvTile, plus 0x100 for the split-K form of a segmented layer: one workgroup per (tile, K
  // segment) writes raw segment sums, splitk_reduce adds them in segment order and applies bias / ReLU --
  // bit-identical to the unsplit launch.  Every output element accumulates its K in the same order whatever the
  // tiling, so codes are bit-neutral.
  auto launch_code = [&](tsm::ConvParams p, int ks, int code) -> hipError_t {
    p.tile = code & 15;
    // A cached code may come from a smaller batch of the same bucket: the scratch-size condition is re-checked on
    // EVERY launch (falling back to the whole-K form, which gives the same bits).
    if ((code & 0x100) && p.kseg_len > 0 &&
        (size_t)tsm::conv_num_segments(p) * (size_t)p.M * (size_t)p.Cout <= e->partial_elems) {
      float *y = p.y;
      p.ksplit = 1;
      p.y = e->d_partial;
      hipError_t st = tsm::launch_conv(p, ks, s);
      if (st == hipSuccess)
        st = tsm::launch_splitk_reduce(e->d_partial, tsm::conv_num_segments(p), p.M, p.Cout, p.bias, nullptr, y, p.relu, s);
      return st;
    }
    // 0x200: the tail split (ConvParams::ksplit = 2) of a segmented 64x64 layer -- the tiles of the last, partly filled round of
    // resident workgroups (five per CU) as (tile, segment) pieces.  Re-derived from THIS launch's tile count on every launch;
    // whenever it does not apply the whole-K form runs (same bits).
    if ((code & 0x200) && (code & 15) == tsm::kTile64x64 && p.kseg_len > 0) {
      int tail_from = 0;
      if (tail_split_from(p, e->n_cu, e->partial_elems, &tail_from)) {
        float *y = p.y;
        const int mt0 = tail_from / (p.Cout / 64) * 64;
        p.ksplit = 2;
        p.tail_from = tail_from;
        p.ypart = e->d_partial;
        hipError_t st = tsm::launch_conv(p, ks, s);
        if (st == hipSuccess)
          st = tsm::launch_splitk_reduce(e->d_partial, tsm::conv_num_segments(p), p.M - mt0, p.Cout, p.bias, nullptr,
                                         y + (size_t)mt0 * p.Cout, p.relu, s);
        return st;
      }
    }
    return tsm::launch_conv(p, ks, s);
  };
  // A code read from TSM_TUNE_CACHE (or forced through the environment) is only trusted after it has been checked
  // against THIS layer: anything else falls back to the heuristic shape, so a stale, foreign or hand-edited cache
  // line can only cost speed.
  auto code_ok = [&](const tsm::ConvParams &p, int code) {
    return code > 0 && (code & ~0x3F0F) == 0 && tsm::conv_tile_valid(p, code & 15);   // (0x400 / 0x800 / 0x1000 / 0x2000: conv2 + conv3 / the whole block / conv3 + the next block's conv1 / conv1 + the stride-2 conv2 run fused, below)
  };
  int flip = 0;   // alternates the tile walk direction of consecutive conv launches (ConvParams::reverse)
  auto conv = [&](int idx, tsm::ConvParams p, int ks, bool is3x3) -> int {
    p.reverse = e->zigzag ? (flip ^= 1) : 0;
    if (!tuning) {
      int code = tiles ? (*tiles)[idx] : 0;
      if (e->force_tile) code = e->force_tile;
      if (e->force_code >= 0) code = e->force_code;   // (launch_code ignores the split bit where it does not apply)
      if (!code_ok(p, code)) code = 0;
      TSM_LAUNCH_K(e, s, is3x3, launch_code(p, ks, code));
      return TSM_OK;
    }
    float best_ms = 0.f;
    int best = 0;
    std::vector<int> cands;
    if (p.kseg_len > 0) {
      // segmented layers: 64x64 / 32x32 tiles, whole-K or split-K (when the segment sums fit the scratch buffer and
      // whole-K tiles alone would leave CUs idle or nearly so)
      const size_t need = (size_t)tsm::conv_num_segments(p) * p.M * p.Cout;
      const long tiles64 = (long)((p.M + 63) / 64) * (p.Cout / 64);
      for (int t : {(int)tsm::kTile64x64, (int)tsm::kTile32x32}) {
        cands.push_back(t);
        if (need <= e->partial_elems && tiles64 < 4L * e->n_cu) cands.push_back(t | 0x100);
      }
      int tail_from = 0;
      if (e->tail_split && tail_split_from(p, e->n_cu, e->partial_elems, &tail_from)) cands.push_back((int)tsm::kTile64x64 | 0x200);
    } else {
      for (int t = 1; t < tsm::kNumTiles; ++t)
        if (tsm::conv_tile_valid(p, t)) cands.push_back(t);
    }
    for (int t : cands) {
      float ms[4];
      for (int rep = 0; rep < 4; ++rep) {
        TSM_HIP(e, hipEventRecord(e->ev0, s));
        TSM_HIP(e, launch_code(p, ks, t));
        TSM_HIP(e, hipEventRecord(e->ev1, s));
        TSM_HIP(e, hipEventSynchronize(e->ev1));
        TSM_HIP(e, hipEventElapsedTime(&ms[rep], e->ev0, e->ev1));
      }
      float m = ms[1] < ms[2] ? ms[1] : ms[2];  // rep 0 warms the caches; best of the other three
      m = ms[3] < m ? ms[3] : m;
      if (best == 0 || m < best_ms) {
        best = t;
        best_ms = m;
      }
    }
    (*tiles)[idx] = best;
    return TSM_OK;
  };

  const int prec = e->prec;
  const float *in4 = e->d_in4;
  // The pool-fused stem reads the reference layout ([N, T, 3, H, W] fp32) itself and converts while it stages its patch:
  // no pack launch, no packed copy (TSM_STEM_PLANAR=0 and every other stem form keep the pack; all bit-identical).
  const bool pool_stem = e->stem_direct && e->stem_pool && !want("conv1");
  // (8-byte loads of pixel pairs: a caller's pointer that is only 4-byte aligned takes the pack launch, which loads floats)
  const bool planar = layout == TSM_LAYOUT_NTCHW && e->stem_planar && pool_stem && !want("input") &&
                      (double)cfg.height * cfg.width * 12.0 < 2.0e9 && (reinterpret_cast<uintptr_t>(d_clips) & 15u) == 0;
  if (layout >= TSM_LAYOUT_NTHWC4 || planar) {
    in4 = d_clips;  // already packed by tsm_preprocess (or read as it is by the stem): consumed in place
    if (e->cur_timing) {  // keep the pack's launch slot: reported as "not recorded"
      e->cur_timing->push_back(nullptr);
      e->cur_timing->push_back(nullptr);
    }
  } else {
    TSM_LAUNCH(e, s, tsm::launch_pack_input(d_clips, e->d_in4, n, cfg.height, cfg.width,
                                            layout == TSM_LAYOUT_NTCHW ? 1 : 0, prec, s));
  }
  if (want("input"))  // fp32: [n,H,W,4]; bf16 formats: pixel pairs [n,H,ceil(W/2),8]
    return prec == tsm::kPrecF32 ? hit(in4, n, cfg.height, cfg.width, 4) : hit(in4, n, cfg.height, (cfg.width + 1) / 2, 8);

  float *cur = e->buf[0], *out = e->buf[1], *t1 = e->buf[2], *t2 = e->buf[3], *idb = e->buf[4];
  {
    tsm::ConvParams p = make_params(e->convs[0], in4, nullptr, t1, n, cfg.height, cfg.width, true, 0, 1, prec);
    // bf16 formats: dedicated direct-conv stem (LDS-resident input patch), with the max-pool fused behind it unless
    // the un-pooled tensor itself is wanted; TSM_STEM_DIRECT=0 / TSM_STEM_POOL=0 fall back (all forms bit-identical)
    const bool direct = prec != tsm::kPrecF32 && e->stem_direct;   // (fp32 has the pool-fused direct form only)
    if (pool_stem) {
      TSM_LAUNCH(e, s, tsm::launch_stem_pool(in4, e->convs[0].d_w, e->convs[0].d_b, cur, n, cfg.height, cfg.width,
                                              e->convs[0].kp, 1, prec, s, planar ? 1 : 0));
      if (e->cur_timing) {  // keep the max-pool's launch slot: reported as "not recorded"
        e->cur_timing->push_back(nullptr);
        e->cur_timing->push_back(nullptr);
      }
    } else {
      if (direct) {
        TSM_LAUNCH(e, s, tsm::launch_stem_direct(in4, e->convs[0].d_w, e->convs[0].d_b, t1, n, cfg.height, cfg.width,
                                                  e->convs[0].kp, 1, prec, s));
      } else {
        int rc0 = conv(0, p, 7, false);
        if (rc0) return rc0;
      }
      if (want("conv1")) return hit(t1, n, e->h1, e->w1, 64);
      TSM_LAUNCH(e, s, tsm::launch_maxpool3x3s2(t1, cur, n, e->h1, e->w1, 64, prec, s));
    }
    if (want("stem")) return hit(cur, n, e->hp, e->wp, 64);
  }
  int h = e->hp, w = e->wp;
  std::vector<std::string> block_names;
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < kBlocks[li]; ++bi) block_names.push_back("layer" + std::to_string(li + 1) + "." + std::to_string(bi));
  bool tapped = false;
  // conv3 of block k and shift + conv1 of block k + 1 as ONE launch (bf16: tsm::launch_conv31_fused): `t1_ready` says that
  // the previous block's conv3 launch has already left this block's conv1 output in t1; `prev3` keeps the previous block's
  // conv3 launch for the tuner's A/B at the head of the next block.
  bool t1_ready = false;
  struct Prev3 {
    bool ok = false;
    tsm::ConvParams p3{};
    int conv3_idx = -1;
  } prev3;
  auto make_p31 = [&](const tsm::ConvParams &p3, const ConvLayer &c1n, float *t1_out, int nn, int hw) {
    tsm::Conv31Params q{};
    q.t2 = p3.x; q.w3 = p3.w; q.bias3 = p3.bias; q.res = p3.res; q.y = p3.y;
    q.w1 = c1n.d_w; q.bias1 = c1n.d_b; q.t1 = t1_out;
    q.n_clips = nn / T; q.T = T; q.HW = hw; q.K3 = p3.C; q.C = p3.Cout; q.N1 = c1n.cout;
    q.fold = shiftT > 0 ? c1n.cp / cfg.shift_div : 0;
    return q;
  };
  // One Bottleneck on nn frames: x -> y through the branch temporaries t1, t2 (and idb for an un-fused downsample).
  auto run_block = [&](size_t k, int nn, float *x, float *y, int hh, int ww) -> int {
    const Block &blk = e->blocks[k];
    const std::string &name = block_names[k];
    const ConvLayer &c1 = e->convs[blk.conv1], &c2 = e->convs[blk.conv2], &c3 = e->convs[blk.conv3];
    const bool have_t1 = t1_ready;
    t1_ready = false;
    const Prev3 prev = prev3;    // (consumed here whatever path this block takes)
    prev3.ok = false;
    const int ho = (hh + 2 - 3) / blk.stride + 1, wo = (ww + 2 - 3) / blk.stride + 1;
    const float *identity = x;
    const bool fused = blk.down >= 0 && blk.d_wf != nullptr;
    if (blk.down >= 0 && !fused) {
      tsm::ConvParams pd = make_params(e->convs[blk.down], x, nullptr, idb, nn, hh, ww, false, 0, 1, prec);
      int rcd = conv(blk.down, pd, 1, false);
      if (rcd) return rcd;
      identity = idb;
    } else if (fused && e->cur_timing) {  // keep the launch slot: reported as "not recorded"
      e->cur_timing->push_back(nullptr);
      e->cur_timing->push_back(nullptr);
    }
    tsm::ConvParams p1 = make_params(c1, x, nullptr, t1, nn, hh, ww, true, shiftT, cfg.shift_div, prec);
    // The WHOLE block as one launch (bf16 layer1: bneck_ws_kernel -- layer1.1 / layer1.2 with the block input as the
    // residual, layer1.0 with the fused conv3 + downsample weights): bit 0x800 of conv1's tile code (set by the tuning
    // pass when it beat conv1 + the best conv2 / conv3 form), or forced / forbidden through TSM_FUSE_BLOCK
    tsm::BneckParams pb{};
    const bool block_shape = prec == tsm::kPrecBf16 && blk.stride == 1 && c1.cout == 64 && c2.cin == 64 && c2.cout == 64 && c3.cout == 256 &&
                             ((blk.down < 0 && c1.cin == 256) || (fused && c1.cin == 64 && e->convs[blk.down >= 0 ? blk.down : 0].cp == 64));
    const bool can_block = block_shape && e->fuse_block != 0 && tsm::bneck_ws_valid(c1.cin, nn, hh, ww, shiftT, p1.fold) &&
                           !want(name + ".conv1") && !want(name + ".conv2");
    if (can_block) {
      pb.x = x; pb.w1 = c1.d_w; pb.bias1 = c1.d_b; pb.w2 = c2.d_w; pb.bias2 = c2.d_b; pb.y = y;
      pb.w3 = fused ? blk.d_wf : c3.d_w; pb.bias3 = fused ? blk.d_bf : c3.d_b;
      pb.cin = c1.cin; pb.N = nn; pb.H = hh; pb.W = ww; pb.T = shiftT; pb.fold = p1.fold;
    }
    if (can_block && !tuning && (e->fuse_block == 1 || (tiles && ((*tiles)[blk.conv1] & 0x800)))) {
      pb.reverse = e->zigzag ? (flip ^= 1) : 0;
      TSM_LAUNCH_K(e, s, false, tsm::launch_bneck_ws(pb, s));
      if (e->cur_timing)   // keep conv2's and conv3's launch slots: reported as "not recorded"
        for (int k = 0; k < 4; ++k) e->cur_timing->push_back(nullptr);
      if (want(name)) { tapped = true; return hit(y, nn, ho, wo, c3.cout); }
      return TSM_OK;
    }
    // shift + conv1 + the stride-2 conv2 of layer2.0 as ONE launch (bf16: front_s2_kernel; t1 never exists in memory): bit 0x2000
    // of conv1's tile code (set by the tuning pass below when it beat the two tuned launches), or forced / forbidden through
    // TSM_FUSE_FRONT
    tsm::FrontParams pfr{};
    const bool can_front = prec == tsm::kPrecBf16 && blk.stride == 2 && c1.cin == 256 && c1.cout == 128 && c2.cin == 128 && c2.cout == 128 &&
                           !have_t1 && e->fuse_front != 0 && tsm::front_s2_valid(nn, hh, ww, shiftT, p1.fold) &&
                           !want(name + ".conv1");
    if (can_front) {
      pfr.x = x; pfr.w1 = c1.d_w; pfr.bias1 = c1.d_b; pfr.w2 = c2.d_w; pfr.bias2 = c2.d_b; pfr.y = t2;
      pfr.N = nn; pfr.H = hh; pfr.W = ww; pfr.T = shiftT; pfr.fold = p1.fold;
    }
    bool did_front = false;
    if (can_front && !tuning && (e->fuse_front == 1 || (tiles && ((*tiles)[blk.conv1] & 0x2000)))) {
      pfr.reverse = e->zigzag ? (flip ^= 1) : 0;
      TSM_LAUNCH_K(e, s, false, tsm::launch_front_s2(pfr, s));
      flip ^= 1;                                    // (it stands for two launches: conv3 keeps the direction it would have had)
      if (e->cur_timing) {                          // keep conv2's launch slot: reported as "not recorded"
        e->cur_timing->push_back(nullptr);
        e->cur_timing->push_back(nullptr);
      }
      did_front = true;
    }
    const int r1 = e->zigzag ? (flip ^ 1) : 0;   // the walk direction conv() is about to give conv1 (kept for the tuner's A/B below)
    if (did_front) {
    } else if (have_t1) {               // the previous block's conv3 launch computed this conv1 as well: keep the launch slot
      if (e->cur_timing) {
        e->cur_timing->push_back(nullptr);
        e->cur_timing->push_back(nullptr);
      }
    } else {
      int rc1 = conv(blk.conv1, p1, 1, false);
      if (rc1) return rc1;
    }
    if (tuning && prev.ok && e->fuse31 < 0) {
      // conv3 of the previous block + this conv1 as one launch against the two tuned launches, same protocol as the other
      // fused forms (both arms rewrite the same bits into the previous block's output and into t1, so re-running is harmless)
      tsm::Conv31Params q = make_p31(prev.p3, c1, t1, nn, hh * ww);
      if (tsm::conv31_valid(q)) {
        const int code3 = (*tiles)[prev.conv3_idx], code1 = (*tiles)[blk.conv1];
        tsm::ConvParams pa = prev.p3, pc = p1;
        pc.reverse = e->zigzag ? (pa.reverse ^ 1) : 0;
        q.reverse = pa.reverse;
        float pair_ms = 0.f, one_ms = 0.f, ms[4];
        for (int arm = 0; arm < 2; ++arm) {
          for (int rep = 0; rep < 4; ++rep) {
            TSM_HIP(e, hipEventRecord(e->ev0, s));
            if (arm == 0) {
              TSM_HIP(e, launch_code(pa, 1, code3));
              TSM_HIP(e, launch_code(pc, 1, code1));
            } else {
              TSM_HIP(e, tsm::launch_conv31_fused(q, s));
            }
            TSM_HIP(e, hipEventRecord(e->ev1, s));
            TSM_HIP(e, hipEventSynchronize(e->ev1));
            TSM_HIP(e, hipEventElapsedTime(&ms[rep], e->ev0, e->ev1));
          }
          (arm == 0 ? pair_ms : one_ms) = std::min(ms[1], std::min(ms[2], ms[3]));
        }
        if (one_ms < pair_ms) (*tiles)[prev.conv3_idx] |= 0x1000;
      }
    }
    if (!did_front && want(name + ".conv1")) { tapped = true; return hit(t1, nn, hh, ww, c1.cout); }
    tsm::ConvParams p2 = make_params(c2, t1, nullptr, t2, nn, hh, ww, true, 0, 1, prec);
    // conv2 + conv3 + residual in one kernel where the block is eligible: bit 0x400 of conv2's tile code (set by the
    // tuning pass when the fused launch beat the two separate ones), or forced / forbidden through TSM_FUSE_CONV23
    tsm::Fused23Params pf{};
    const bool ws23 = prec == tsm::kPrecBf16 && blk.cmid == 64 && tsm::conv23_ws_valid(nn, hh, ww);
    const bool can_fuse = (blk.d_w3f != nullptr || ws23) && e->fuse23 != 0 && !want(name + ".conv2");
    if (can_fuse) {
      pf.x = t1; pf.w2 = c2.d_w; pf.bias2 = c2.d_b; pf.w3f = ws23 ? c3.d_w : blk.d_w3f; pf.bias3 = c3.d_b; pf.res = identity; pf.y = y;
      pf.N = nn; pf.H = hh; pf.W = ww; pf.M = nn * hh * ww; pf.kseg_len = c2.kseg;
      pf.reverse = e->zigzag ? (flip ^ 1) : 0;
    }
    if (can_fuse && !tuning && (e->fuse23 == 1 || (tiles && ((*tiles)[blk.conv2] & 0x400)))) {
      TSM_LAUNCH_K(e, s, true, tsm::launch_conv23_fused(pf, blk.cmid, prec, s));
      flip ^= 1;
      if (e->cur_timing) {  // keep conv3's launch slot: reported as "not recorded"
        e->cur_timing->push_back(nullptr);
        e->cur_timing->push_back(nullptr);
      }
      if (want(name)) { tapped = true; return hit(y, nn, ho, wo, c3.cout); }
      return TSM_OK;
    }
    if (!did_front) {
      int rc2 = conv(blk.conv2, p2, 3, true);
      if (rc2) return rc2;
      if (tuning && can_front && e->fuse_front < 0) {
        // the one launch against the two tuned ones, same protocol as the other fused forms (both arms write the same bits to t2)
        const int code1 = (*tiles)[blk.conv1], code2 = (*tiles)[blk.conv2];
        tsm::ConvParams pa = p1, pc = p2;
        pa.reverse = r1;
        pc.reverse = e->zigzag ? (r1 ^ 1) : 0;
        pfr.reverse = r1;
        float pair_ms = 0.f, one_ms = 0.f, ms[4];
        for (int arm = 0; arm < 2; ++arm) {
          for (int rep = 0; rep < 4; ++rep) {
            TSM_HIP(e, hipEventRecord(e->ev0, s));
            if (arm == 0) {
              TSM_HIP(e, launch_code(pa, 1, code1));
              TSM_HIP(e, launch_code(pc, 3, code2));
            } else {
              TSM_HIP(e, tsm::launch_front_s2(pfr, s));
            }
            TSM_HIP(e, hipEventRecord(e->ev1, s));
            TSM_HIP(e, hipEventSynchronize(e->ev1));
            TSM_HIP(e, hipEventElapsedTime(&ms[rep], e->ev0, e->ev1));
          }
          (arm == 0 ? pair_ms : one_ms) = std::min(ms[1], std::min(ms[2], ms[3]));
        }
        if (one_ms < pair_ms) (*tiles)[blk.conv1] |= 0x2000;
      }
    }
    if (want(name + ".conv2")) { tapped = true; return hit(t2, nn, ho, wo, c2.cout); }
    tsm::ConvParams p3 = make_params(c3, t2, fused ? nullptr : identity, y, nn, ho, wo, true, 0, 1, prec);
    if (fused) {
      const ConvLayer &cd = e->convs[blk.down];
      p3.w = blk.d_wf; p3.bias = blk.d_bf; p3.Kp = blk.kpf; p3.K1 = c3.kp; p3.kseg_len = blk.ksegf;
      p3.x2 = x; p3.C2 = cd.cp; p3.Hi2 = hh; p3.Wi2 = ww; p3.stride2 = blk.stride;
    }
    // conv3 + residual of this block and shift + conv1 of the NEXT block as one launch (bf16; this block without a downsample
    // branch -- the next one may be the first of a stage: its conv1 is a stride-1 1x1 on this block's output either way):
    // bit 0x1000 of conv3's tile code
    // (set by the tuning pass at the head of the next block), or forced / forbidden through TSM_FUSE_C3C1
    bool did31 = false;
    if (prec == tsm::kPrecBf16 && !fused && blk.down < 0 && k + 1 < e->blocks.size() && e->fuse31 != 0) {
      const ConvLayer &c1n = e->convs[e->blocks[k + 1].conv1];
      tsm::Conv31Params q = make_p31(p3, c1n, t1, nn, ho * wo);
      if (tsm::conv31_valid(q)) {
        if (tuning) {
          prev3.ok = true;
          prev3.conv3_idx = blk.conv3;
        } else if (e->fuse31 == 1 || (tiles && ((*tiles)[blk.conv3] & 0x1000))) {
          q.reverse = e->zigzag ? (flip ^= 1) : 0;
          TSM_LAUNCH_K(e, s, false, tsm::launch_conv31_fused(q, s));
          did31 = true;
          t1_ready = true;
        }
      }
    }
    if (!did31) {
      const int r3 = e->zigzag ? (flip ^ 1) : 0;
      int rc3 = conv(blk.conv3, p3, 1, false);
      if (rc3) return rc3;
      if (prev3.ok) {
        prev3.p3 = p3;
        prev3.p3.reverse = r3;
      }
    }
    if (tuning && ((can_fuse && e->fuse23 < 0) || (can_block && e->fuse_block < 0))) {
      // Fused forms against the tuned separate launches (same bits), under ONE protocol: four repetitions, the first
      // discarded, best of the other three, and a sequence of launches timed back to back inside one event bracket (so
      // the inter-kernel boundaries and each kernel's cold start on its producer's output are priced, as in a forward).
      const int code1 = (*tiles)[blk.conv1], code2 = (*tiles)[blk.conv2], code3 = (*tiles)[blk.conv3];
      // ... and with the tile walk directions of a forward (conv() sets `reverse` on its own copy only): separate launches
      // alternate r1, !r1, r1; the fused conv2 + conv3 walks like conv2, the whole block like conv1 -- both arms of every
      // comparison then see the producer / consumer cache reuse they would get in a real forward
      p1.reverse = r1;
      p2.reverse = e->zigzag ? (r1 ^ 1) : 0;
      p3.reverse = r1;
      pf.reverse = p2.reverse;
      pb.reverse = r1;
      auto best_of = [&](auto &&launch, float *out_ms) -> int {
        float ms[4];
        for (int rep = 0; rep < 4; ++rep) {
          TSM_HIP(e, hipEventRecord(e->ev0, s));
          TSM_HIP(e, launch());
          TSM_HIP(e, hipEventRecord(e->ev1, s));
          TSM_HIP(e, hipEventSynchronize(e->ev1));
          TSM_HIP(e, hipEventElapsedTime(&ms[rep], e->ev0, e->ev1));
        }
        *out_ms = std::min(ms[1], std::min(ms[2], ms[3]));
        return TSM_OK;
      };
      auto pair = [&]() -> hipError_t {
        hipError_t st = launch_code(p2, 3, code2);
        return st != hipSuccess ? st : launch_code(p3, 1, code3);
      };
      bool use_fused = can_fuse && e->fuse23 == 1;
      if (can_fuse && e->fuse23 < 0) {       // conv2 + conv3 as one kernel against the pair
        float pair_ms = 0.f, fused_ms = 0.f;
        int rcp = best_of(pair, &pair_ms);
        if (rcp) return rcp;
        rcp = best_of([&]() -> hipError_t { return tsm::launch_conv23_fused(pf, blk.cmid, prec, s); }, &fused_ms);
        if (rcp) return rcp;
        use_fused = fused_ms < pair_ms;
        if (use_fused) (*tiles)[blk.conv2] |= 0x400;
      }
      if (can_block && e->fuse_block < 0) {   // the whole block as one kernel against conv1 + the better form of the rest
        float rest_ms = 0.f, block_ms = 0.f;
        int rcp = best_of([&]() -> hipError_t {
          hipError_t st = launch_code(p1, 1, code1);
          if (st != hipSuccess) return st;
          return use_fused ? tsm::launch_conv23_fused(pf, blk.cmid, prec, s) : pair();
        }, &rest_ms);
        if (rcp) return rcp;
        rcp = best_of([&]() -> hipError_t { return tsm::launch_bneck_ws(pb, s); }, &block_ms);
        if (rcp) return rcp;
        if (block_ms < rest_ms) (*tiles)[blk.conv1] |= 0x800;
      }
    }
    if (want(name)) { tapped = true; return hit(y, nn, ho, wo, c3.cout); }
    return TSM_OK;
  };
  for (size_t k = 0; k < e->blocks.size(); ++k) {
    int rc = run_block(k, n, cur, out, h, w);
    if (rc || tapped) return rc;
    std::swap(cur, out);
    h = (h + 2 - 3) / e->blocks[k].stride + 1;
    w = (w + 2 - 3) / e->blocks[k].stride + 1;
  }
  if (stage) return fail(e, TSM_ERR_INVALID_ARG, std::string("unknown stage: ") + stage);
  TSM_LAUNCH(e, s, tsm::launch_head(cur, e->d_fcw, e->d_fcb, e->d_pooled, d_logits, n_clips, T, h * w, 2048,
                                    cfg.num_class, prec, s));
  return TSM_OK;
}

// The tile codes of the bucket `n_clips` falls into: already in memory, read from the tune cache file, or timed now on
// the real launches over `d_clips` (synchronises; the codes then go to the cache file).
int ensure_tuned(tsm_engine *e, const float *d_clips, int layout, int n_clips, float *d_out, hipStream_t s,
                 bool *from_file = nullptr) {
  const int tune_key = tile_bucket(n_clips) * e->cfg.num_segments;
  if (from_file) *from_file = false;
  if (!e->autotune || e->tile_cache.find(tune_key) != e->tile_cache.end()) return TSM_OK;
  std::vector<int> cached(e->convs.size(), 0);
  if (tune_cache_load(e, tune_key, &cached)) {
    e->tile_cache.emplace(tune_key, cached);          // tuned by an earlier process (TSM_TUNE_CACHE)
    if (from_file) *from_file = true;
    return TSM_OK;
  }
  std::vector<hipEvent_t> *saved = e->cur_timing;
  e->cur_timing = nullptr;
  const int rc = run_forward(e, d_clips, layout, n_clips, d_out, s, nullptr, nullptr);
  e->cur_timing = saved;
  if (rc) return rc;
  tune_cache_store(e, tune_key, e->tile_cache[tune_key]);
  return TSM_OK;
}

// Device-memory calls run where the caller's data lives: `stream`, NULL meaning the device's default
// (null) stream -- which is what torch's default stream is.  Host-memory calls are synchronous and use
// the engine's private stream unless one is given.
hipStream_t pick_stream(tsm_engine *e, int memkind, void *stream) {
  if (stream) return static_cast<hipStream_t>(stream);
  return memkind == TSM_MEM_HOST ? e->stream : nullptr;
}

int check_forward_args(tsm_engine *e, const void *clips, int memkind, int layout, int n_clips) {
  if (!e) return TSM_ERR_INVALID_ARG;
  if (!e->finalized) return fail(e, TSM_ERR_NOT_FINALIZED, "tsm_finalize has not been called");
  if (!clips) return fail(e, TSM_ERR_INVALID_ARG, "clips is NULL");
  if (memkind != TSM_MEM_HOST && memkind != TSM_MEM_DEVICE) return fail(e, TSM_ERR_INVALID_ARG, "bad memkind");
  if (layout < TSM_LAYOUT_NTCHW || layout > TSM_LAYOUT_NTHWC8B) return fail(e, TSM_ERR_INVALID_ARG, "bad layout");
  if (layout >= TSM_LAYOUT_NTHWC4 && memkind != TSM_MEM_DEVICE)
    return fail(e, TSM_ERR_INVALID_ARG, "packed layouts (NTHWC4 / NTHWC8S / NTHWC8B) are device-memory layouts");
  if ((layout == TSM_LAYOUT_NTHWC4 && e->prec != tsm::kPrecF32) ||
      (layout == TSM_LAYOUT_NTHWC8S && e->prec != tsm::kPrecBf16x3) ||
      (layout == TSM_LAYOUT_NTHWC8B && e->prec != tsm::kPrecBf16))
    return fail(e, TSM_ERR_INVALID_ARG, "packed layout does not match the engine dtype");
  if (n_clips <= 0) return fail(e, TSM_ERR_INVALID_ARG, "n_clips must be positive");
  if (n_clips > e->cfg.max_clips)
    return fail(e, TSM_ERR_CAPACITY, "n_clips " + std::to_string(n_clips) + " exceeds max_clips " +
                                         std::to_string(e->cfg.max_clips));
  return TSM_OK;
}

}  // namespace

extern "C" {

int tsm_abi_version(void) { return TSM_ABI_VERSION; }

const char *tsm_build_id(void) { return kBuildTag + sizeof("tsm-build-id:") - 1; }

int tsm_trace_launches(int32_t on) {
  tsm::trace_launches(on != 0);
  return TSM_OK;
}

int64_t tsm_launch_trace(char *buf, int64_t cap) {
  const char *t = tsm::launch_trace();
  const int64_t need = (int64_t)strlen(t) + 1;
  if (buf && cap >= need) memcpy(buf, t, (size_t)need);
  return need;
}

const char *tsm_last_error(const tsm_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int tsm_create(const tsm_config *cfg, tsm_engine **out) {
  if (!cfg || !out) return fail(nullptr, TSM_ERR_INVALID_ARG, "cfg/out is NULL");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(tsm_config))
    return fail(nullptr, TSM_ERR_INVALID_ARG, "tsm_config.struct_size mismatch (ABI)");
  if (cfg->num_class <= 0 || cfg->num_segments <= 0 || cfg->max_clips <= 0)
    return fail(nullptr, TSM_ERR_INVALID_ARG, "num_class, num_segments, max_clips must be positive");
  if (cfg->height < 32 || cfg->width < 32)
    return fail(nullptr, TSM_ERR_INVALID_ARG, "height/width must be >= 32");
  {  // the conv kernels index output rows (frames x stem output pixels) with 32-bit ints
    const int64_t rows = (int64_t)cfg->max_clips * cfg->num_segments * ((cfg->height + 1) / 2) * ((cfg->width + 1) / 2);
    if (rows >= (int64_t)1 << 31)
      return fail(nullptr, TSM_ERR_CAPACITY, "max_clips * num_segments * (H/2) * (W/2) must stay below 2^31");
  }
  if (cfg->shift_div <= 0 || (64 % cfg->shift_div) != 0 || (64 / cfg->shift_div) % 4 != 0)
    return fail(nullptr, TSM_ERR_UNSUPPORTED, "shift_div must divide 64 with fold % 4 == 0 (8 or 16... )");
  if (cfg->dtype != TSM_DTYPE_F32 && cfg->dtype != TSM_DTYPE_BF16X3 && cfg->dtype != TSM_DTYPE_BF16)
    return fail(nullptr, TSM_ERR_UNSUPPORTED, "dtype must be TSM_DTYPE_F32, TSM_DTYPE_BF16X3 or TSM_DTYPE_BF16");
  if (cfg->dtype != TSM_DTYPE_F32 && (64 / cfg->shift_div) % 8 != 0)
    return fail(nullptr, TSM_ERR_UNSUPPORTED, "the bf16 formats need fold % 8 == 0 (shift_div <= 8)");
  int ndev = 0;
  hipError_t st = hipGetDeviceCount(&ndev);
  if (st != hipSuccess || ndev <= 0)
    return fail(nullptr, TSM_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(st));
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, TSM_ERR_INVALID_ARG, "bad device_id");
  tsm_engine *e = new tsm_engine();
  e->cfg = *cfg;
  e->prec = cfg->dtype == TSM_DTYPE_BF16X3 ? tsm::kPrecBf16x3
            : cfg->dtype == TSM_DTYPE_BF16 ? tsm::kPrecBf16 : tsm::kPrecF32;
  if (const char *at = getenv("TSM_AUTOTUNE")) e->autotune = atoi(at) != 0;
  if (const char *fd = getenv("TSM_FUSE_DOWNSAMPLE")) e->fuse_down = atoi(fd) != 0;
  if (const char *sd = getenv("TSM_STEM_DIRECT")) e->stem_direct = atoi(sd) != 0;
  if (const char *sp = getenv("TSM_STEM_POOL")) e->stem_pool = atoi(sp) != 0;
  if (const char *pl = getenv("TSM_STEM_PLANAR")) e->stem_planar = atoi(pl) != 0;
  if (const char *ft = getenv("TSM_CONV_TILE")) e->force_tile = tsm::conv_tile_from_name(ft);
  if (const char *fc = getenv("TSM_CONV_CODE")) e->force_code = atoi(fc);
  if (const char *f23 = getenv("TSM_FUSE_CONV23")) e->fuse23 = atoi(f23) != 0;
  if (const char *fb = getenv("TSM_FUSE_BLOCK")) e->fuse_block = atoi(fb) != 0;
  if (const char *f31 = getenv("TSM_FUSE_C3C1")) e->fuse31 = atoi(f31) != 0;
  if (const char *ff = getenv("TSM_FUSE_FRONT")) e->fuse_front = atoi(ff) != 0;
  if (const char *zz = getenv("TSM_ZIGZAG")) e->zigzag = atoi(zz) != 0;
  if (const char *ts = getenv("TSM_TAIL_SPLIT")) e->tail_split = atoi(ts) != 0;
  // TSM_TUNE_CACHE=<file> names the tune cache; unset: a per-user default ($XDG_CACHE_HOME or $HOME/.cache, then
  // tsm_hip/tune_cache.txt), so that the second process on a machine pays no tuning pass; "", "0" or "off" disables it.
  {
    const char *tc = getenv("TSM_TUNE_CACHE");
    if (tc) {
      if (*tc && strcmp(tc, "0") != 0 && strcmp(tc, "off") != 0) e->tune_path = tc;
    } else {
      e->tune_path = default_tune_cache_path();
    }
  }
  build_topology(e);
  st = hipSetDevice(cfg->device_id);
  if (st == hipSuccess) st = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (st == hipSuccess) st = hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device_id);
  if (st == hipSuccess) st = hipEventCreate(&e->ev0);
  if (st == hipSuccess) st = hipEventCreate(&e->ev1);
  if (st != hipSuccess) {
    g_create_error = std::string("engine init: ") + hipGetErrorString(st);
    delete e;
    return TSM_ERR_HIP;
  }
  if (!e->tune_path.empty()) {
    hipDeviceProp_t prop;
    // (the build id -- csrc/, per-file options AND the build's extra definitions -- : codes of another build of the library
    //  are still SAFE, every code is validated against its layer at launch, but they were timed on other kernels, so they
    //  are not reused; the same goes for everything else that changes the timings: CU count, walk order, stem forms)
    e->tune_sig = "abi" + std::to_string(TSM_ABI_VERSION) + " build " TSM_BUILD_ID " " +
                  (hipGetDeviceProperties(&prop, cfg->device_id) == hipSuccess ? std::string(prop.gcnArchName) : "?") +
                  " cu" + std::to_string(e->n_cu) +
                  " T" + std::to_string(cfg->num_segments) + " " + std::to_string(cfg->height) + "x" +
                  std::to_string(cfg->width) + " dtype" + std::to_string(cfg->dtype) + " shift" +
                  std::to_string(cfg->is_shift ? cfg->shift_div : 0) + " fuse" + std::to_string(e->fuse_down ? 1 : 0) + "/" + std::to_string(e->fuse23) + "/" + std::to_string(e->fuse_block) + "/" + std::to_string(e->fuse31) + "/" + std::to_string(e->fuse_front) +
                  " zz" + std::to_string(e->zigzag ? 1 : 0) + " tk" + std::to_string(e->tail_split ? 1 : 0) + " stem" + std::to_string(e->stem_direct ? 1 : 0) +
                  std::to_string(e->stem_pool ? 1 : 0) + std::to_string(e->stem_planar ? 1 : 0);
  }
  *out = e;
  return TSM_OK;
}

void tsm_destroy(tsm_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device_id);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (void *p : e->allocs) (void)hipFree(p);
  for (auto &v : e->timing)
    for (hipEvent_t ev : v)
      if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int tsm_set_tensor(tsm_engine *e, const char *name, const float *host_data, const int64_t *shape,
                   int32_t ndim) {
  if (!e) return TSM_ERR_INVALID_ARG;
  if (!name || !host_data || !shape || ndim < 1 || ndim > 4) return fail(e, TSM_ERR_INVALID_ARG, "bad tensor args");
  if (e->finalized) return fail(e, TSM_ERR_INVALID_ARG, "engine already finalized");
  const std::string key(name);
  if (!known_name(e, key)) return fail(e, TSM_ERR_INVALID_ARG, "unknown tensor name: " + key);
  HostTensor t;
  size_t elems = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return fail(e, TSM_ERR_SHAPE, "non-positive dim in " + key);
    t.shape.push_back(shape[i]);
    elems *= (size_t)shape[i];
  }
  t.data.assign(host_data, host_data + elems);
  e->tensors[key] = std::move(t);
  return TSM_OK;
}

int tsm_finalize(tsm_engine *e) {
  if (!e) return TSM_ERR_INVALID_ARG;
  if (e->finalized) return TSM_OK;
  TSM_HIP(e, hipSetDevice(e->cfg.device_id));
  const tsm_config &cfg = e->cfg;
  std::vector<std::vector<float>> host_wp(e->convs.size()), host_bias(e->convs.size());
  for (size_t ci = 0; ci < e->convs.size(); ++ci) {
    ConvLayer &c = e->convs[ci];
    const HostTensor *w = find_tensor(e, c.wkey);
    const HostTensor *g = find_tensor(e, c.bnp + ".weight"), *b = find_tensor(e, c.bnp + ".bias");
    const HostTensor *m = find_tensor(e, c.bnp + ".running_mean"), *v = find_tensor(e, c.bnp + ".running_var");
    if (!w) return fail(e, TSM_ERR_MISSING_TENSOR, "missing " + c.wkey);
    if (!g || !b || !m || !v) return fail(e, TSM_ERR_MISSING_TENSOR, "missing BatchNorm tensors of " + c.bnp);
    const std::vector<int64_t> want = {c.cout, c.cin, c.k, c.k};
    if (w->shape != want) return fail(e, TSM_ERR_SHAPE, "shape mismatch for " + c.wkey);
    for (const HostTensor *t : {g, b, m, v})
      if (t->shape.size() != 1 || t->shape[0] != c.cout) return fail(e, TSM_ERR_SHAPE, "BN shape mismatch for " + c.bnp);
    std::vector<float> wp, bias;
    if (c.k == 7 && e->prec != tsm::kPrecF32)
      fold_and_pack_stem_pairs(w->data.data(), g->data.data(), b->data.data(), m->data.data(), v->data.data(), c.cout,
                               c.kp, &wp, &bias);
    else
      fold_and_pack(w->data.data(), g->data.data(), b->data.data(), m->data.data(), v->data.data(), c.cout,
                    c.cin, c.k, c.cp, c.kp, &wp, &bias);
    if (c.k == 1) {  // fp32 packed copies of the 1x1 layers, for the conv3 + downsample fusion below
      host_wp[ci] = wp;
      host_bias[ci] = bias;
    }
    if (e->prec == tsm::kPrecBf16x3) to_split(&wp);
    if (e->prec == tsm::kPrecBf16) to_bf16(&wp);
    int rc = dev_alloc(e, &c.d_w, wp.size());
    if (rc) return rc;
    rc = dev_alloc(e, &c.d_b, bias.size());
    if (rc) return rc;
    TSM_HIP(e, hipMemcpy(c.d_w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
    TSM_HIP(e, hipMemcpy(c.d_b, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // First block of every stage: out = relu(conv3(h2) + downsample(x)) as one GEMM, K concatenated.
  for (Block &blk : e->blocks) {
    if (blk.down < 0 || !e->fuse_down) continue;
    const ConvLayer &c3 = e->convs[blk.conv3], &cd = e->convs[blk.down];
    blk.kpf = c3.kp + cd.kp;
    blk.ksegf = segment_len(blk.kpf, e->prec);
    std::vector<float> wf((size_t)c3.cout * blk.kpf), bf(c3.cout);
    for (int o = 0; o < c3.cout; ++o) {
      std::memcpy(&wf[(size_t)o * blk.kpf], &host_wp[blk.conv3][(size_t)o * c3.kp], c3.kp * sizeof(float));
      std::memcpy(&wf[(size_t)o * blk.kpf + c3.kp], &host_wp[blk.down][(size_t)o * cd.kp], cd.kp * sizeof(float));
      bf[o] = host_bias[blk.conv3][o] + host_bias[blk.down][o];
    }
    if (e->prec == tsm::kPrecBf16x3) to_split(&wf);
    if (e->prec == tsm::kPrecBf16) to_bf16(&wf);
    int rcf = dev_alloc(e, &blk.d_wf, wf.size());
    if (rcf) return rcf;
    rcf = dev_alloc(e, &blk.d_bf, bf.size());
    if (rcf) return rcf;
    TSM_HIP(e, hipMemcpy(blk.d_wf, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice));
    TSM_HIP(e, hipMemcpy(blk.d_bf, bf.data(), bf.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // Blocks without a downsample branch whose mid tensor is 64 / 128 channels wide: conv3's weights once more, in the
  // fragment order of the fused conv2 + conv3 kernel (fp32 and split-bf16 engines).
  for (Block &blk : e->blocks) {
    const ConvLayer &c2 = e->convs[blk.conv2], &c3 = e->convs[blk.conv3];
    if (blk.down >= 0 || blk.stride != 1 || (c2.cout != 64 && c2.cout != 128)) continue;
    if (e->prec == tsm::kPrecBf16) {   // weight-stationary form: 64 mid channels only; it reads conv3's packed matrix itself
      if (c2.cout == 64 && c2.cin == 64 && c3.cout == 256) blk.cmid = 64;
      continue;
    }
    std::vector<float> w3f;
    if (e->prec == tsm::kPrecBf16x3) pack_w3_fragments_split(host_wp[blk.conv3].data(), c2.cout, &w3f);
    else pack_w3_fragments(host_wp[blk.conv3].data(), c2.cout, &w3f);
    int rcw = dev_alloc(e, &blk.d_w3f, w3f.size());
    if (rcw) return rcw;
    TSM_HIP(e, hipMemcpy(blk.d_w3f, w3f.data(), w3f.size() * sizeof(float), hipMemcpyHostToDevice));
    blk.cmid = c2.cout;
    (void)c3;
  }
  host_wp.clear();
  host_bias.clear();
  const HostTensor *fw = find_tensor(e, "fc.weight"), *fb = find_tensor(e, "fc.bias");
  if (!fw || !fb) return fail(e, TSM_ERR_MISSING_TENSOR, "missing fc.weight / fc.bias");
  if (fw->shape != std::vector<int64_t>{cfg.num_class, 2048} || fb->shape != std::vector<int64_t>{cfg.num_class})
    return fail(e, TSM_ERR_SHAPE, "fc shape mismatch (want [num_class, 2048])");
  int rc = dev_alloc(e, &e->d_fcw, fw->data.size());
  if (rc) return rc;
  rc = dev_alloc(e, &e->d_fcb, fb->data.size());
  if (rc) return rc;
  TSM_HIP(e, hipMemcpy(e->d_fcw, fw->data.data(), fw->data.size() * sizeof(float), hipMemcpyHostToDevice));
  TSM_HIP(e, hipMemcpy(e->d_fcb, fb->data.data(), fb->data.size() * sizeof(float), hipMemcpyHostToDevice));

  // Workspace: the largest activation is the stem conv output (== layer1 output), per frame
  // h1*w1*64 floats; five rotating buffers (block in/out, two branch temporaries, identity).
  e->h1 = (cfg.height + 6 - 7) / 2 + 1;
  e->w1 = (cfg.width + 6 - 7) / 2 + 1;
  e->hp = (e->h1 + 2 - 3) / 2 + 1;
  e->wp = (e->w1 + 2 - 3) / 2 + 1;
  const size_t frames = (size_t)cfg.max_clips * cfg.num_segments;
  size_t per_frame = (size_t)e->h1 * e->w1 * 64;
  const size_t l1 = (size_t)e->hp * e->wp * 256;
  if (l1 > per_frame) per_frame = l1;
  e->buf_elems = frames * per_frame;
  for (int i = 0; i < 5; ++i) {
    rc = dev_alloc(e, &e->buf[i], e->buf_elems);
    if (rc) return rc;
  }
  rc = dev_alloc(e, &e->d_in, frames * 3 * cfg.height * cfg.width);
  if (rc) return rc;
  rc = dev_alloc(e, &e->d_in4, frames * 4 * cfg.height * (cfg.width + 1));  // 16 bytes per pixel in every format (pairs: odd widths padded)
  if (rc) return rc;
  rc = dev_alloc(e, &e->d_pooled, frames * 2048);
  if (rc) return rc;
  rc = dev_alloc(e, &e->d_logits, (size_t)cfg.max_clips * cfg.num_class);
  if (rc) return rc;
  if (e->prec == tsm::kPrecF32) {  // split-K scratch: 64 MB covers the small-batch cases where split-K can win
    e->partial_elems = (size_t)16 << 20;
    rc = dev_alloc(e, &e->d_partial, e->partial_elems);
    if (rc) return rc;
  }
  e->tensors.clear();  // host copies are no longer needed
  e->finalized = true;
  return TSM_OK;
}

int tsm_forward(tsm_engine *e, const void *clips, int32_t memkind, int32_t layout, int32_t n_clips,
                float *logits, void *stream) {
  int rc = check_forward_args(e, clips, memkind, layout, n_clips);
  if (rc) return rc;
  if (!logits) return fail(e, TSM_ERR_INVALID_ARG, "logits is NULL");
  TSM_HIP(e, hipSetDevice(e->cfg.device_id));
  hipStream_t s = pick_stream(e, memkind, stream);
  const size_t in_elems = (size_t)n_clips * e->cfg.num_segments * 3 * e->cfg.height * e->cfg.width;
  const float *d_clips = static_cast<const float *>(clips);
  float *d_out = logits;
  if (memkind == TSM_MEM_HOST) {
    TSM_HIP(e, hipMemcpyAsync(e->d_in, clips, in_elems * sizeof(float), hipMemcpyHostToDevice, s));
    d_clips = e->d_in;
    d_out = e->d_logits;
  }
  e->cur_timing = nullptr;
  if (e->timing_left > 0) {
    --e->timing_left;
    e->timing.emplace_back();
    e->cur_timing = &e->timing.back();
  }
  // A forward that still has to tune its tile shapes does so in a throw-away pass first (it uses the
  // timing events and synchronises); the real pass below then runs from the cache.
  rc = ensure_tuned(e, d_clips, layout, n_clips, d_out, s);
  if (rc) return rc;
  TSM_HIP(e, hipEventRecord(e->ev0, s));
  rc = run_forward(e, d_clips, layout, n_clips, d_out, s, nullptr, nullptr);
  e->cur_timing = nullptr;
  if (rc) return rc;
  TSM_HIP(e, hipEventRecord(e->ev1, s));
  e->have_time = true;
  if (memkind == TSM_MEM_HOST) {
    TSM_HIP(e, hipMemcpyAsync(logits, e->d_logits, (size_t)n_clips * e->cfg.num_class * sizeof(float),
                              hipMemcpyDeviceToHost, s));
    TSM_HIP(e, hipStreamSynchronize(s));
  }
  return TSM_OK;
}

int tsm_tune(tsm_engine *e, int32_t n_clips, void *stream) {
  if (!e) return TSM_ERR_INVALID_ARG;
  if (!e->finalized) return fail(e, TSM_ERR_NOT_FINALIZED, "tsm_finalize has not been called");
  if (n_clips <= 0) return fail(e, TSM_ERR_INVALID_ARG, "n_clips must be positive");
  if (n_clips > e->cfg.max_clips)
    return fail(e, TSM_ERR_CAPACITY, "n_clips " + std::to_string(n_clips) + " exceeds max_clips " + std::to_string(e->cfg.max_clips));
  const int tune_key = tile_bucket(n_clips) * e->cfg.num_segments;
  if (!e->autotune || e->tile_cache.find(tune_key) != e->tile_cache.end()) return TSM_OK;
  TSM_HIP(e, hipSetDevice(e->cfg.device_id));
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the engine's own packed-input buffer, zeroed, stands in for a batch: no caller memory, no host copy, no kernel of
  // anybody else's (a cold process pays first-use code-object loads for those)
  const size_t frames = (size_t)n_clips * e->cfg.num_segments;
  TSM_HIP(e, hipMemsetAsync(e->d_in4, 0, frames * 4 * e->cfg.height * (e->cfg.width + 1) * sizeof(float), s));
  const int layout = e->prec == tsm::kPrecF32 ? TSM_LAYOUT_NTHWC4 : e->prec == tsm::kPrecBf16x3 ? TSM_LAYOUT_NTHWC8S : TSM_LAYOUT_NTHWC8B;
  bool from_file = false;
  int rc = ensure_tuned(e, e->d_in4, layout, n_clips, e->d_logits, s, &from_file);
  if (rc || !from_file) return rc;
  // The choices came from the cache file, so nothing has been launched yet: run the bucket's forward once on the zeroed
  // buffer anyway.  Every kernel of the schedule is then loaded (the library bundles one code object per kernel family,
  // each loaded lazily at its first launch) and the header's promise holds on this path too -- after tsm_tune a
  // tsm_forward of the bucket allocates nothing, loads nothing and synchronises nothing.
  std::vector<hipEvent_t> *saved = e->cur_timing;
  e->cur_timing = nullptr;
  rc = run_forward(e, e->d_in4, layout, n_clips, e->d_logits, s, nullptr, nullptr);
  e->cur_timing = saved;
  return rc;
}

int tsm_forward_tap(tsm_engine *e, const void *clips, int32_t memkind, int32_t layout, int32_t n_clips,
                    const char *stage, float *out, int64_t out_capacity, int64_t out_shape[4],
                    void *stream) {
  int rc = check_forward_args(e, clips, memkind, layout, n_clips);
  if (rc) return rc;
  if (!stage || !out || !out_shape) return fail(e, TSM_ERR_INVALID_ARG, "stage/out/out_shape is NULL");
  TSM_HIP(e, hipSetDevice(e->cfg.device_id));
  hipStream_t s = pick_stream(e, memkind, stream);
  const size_t in_elems = (size_t)n_clips * e->cfg.num_segments * 3 * e->cfg.height * e->cfg.width;
  const float *d_clips = static_cast<const float *>(clips);
  if (memkind == TSM_MEM_HOST) {
    TSM_HIP(e, hipMemcpyAsync(e->d_in, clips, in_elems * sizeof(float), hipMemcpyHostToDevice, s));
    d_clips = e->d_in;
  }
  Tap tap;
  rc = run_forward(e, d_clips, layout, n_clips, e->d_logits, s, stage, &tap);
  if (rc) return rc;
  const int64_t elems = tap.shape[0] * tap.shape[1] * tap.shape[2] * tap.shape[3];
  for (int i = 0; i < 4; ++i) out_shape[i] = tap.shape[i];
  if (elems > out_capacity) return fail(e, TSM_ERR_CAPACITY, "tap output buffer too small");
  if (e->prec != tsm::kPrecF32) {  // taps are reported as fp32 whatever the storage format
    if (!e->d_tap) {
      const size_t cap = e->buf_elems > (size_t)e->cfg.max_clips * e->cfg.num_segments * 8 * e->cfg.height * e->cfg.width
                             ? e->buf_elems
                             : (size_t)e->cfg.max_clips * e->cfg.num_segments * 8 * e->cfg.height * e->cfg.width;
      rc = dev_alloc(e, &e->d_tap, cap);
      if (rc) return rc;
    }
    TSM_HIP(e, tsm::launch_to_f32(tap.ptr, e->d_tap, elems / 8, e->prec, s));
    tap.ptr = e->d_tap;
  }
  TSM_HIP(e, hipMemcpyAsync(out, tap.ptr, (size_t)elems * sizeof(float),
                            memkind == TSM_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
  TSM_HIP(e, hipStreamSynchronize(s));
  return TSM_OK;
}

int tsm_set_layer_timing(tsm_engine *e, int32_t n_forwards, int32_t only_conv3x3) {
  if (!e) return TSM_ERR_INVALID_ARG;
  if (n_forwards < 0 || n_forwards > 64) return fail(e, TSM_ERR_INVALID_ARG, "n_forwards must be in [0, 64]");
  e->timing_only3x3 = only_conv3x3 != 0;
  for (auto &v : e->timing)
    for (hipEvent_t ev : v)
      if (ev) e->event_pool.push_back(ev);
  e->timing.clear();
  e->timing.reserve(64);
  e->timing_left = n_forwards;
  return TSM_OK;
}

int tsm_layer_times(tsm_engine *e, int32_t forward_index, float *ms_out, int32_t cap, int32_t *n_out) {
  if (!e || !ms_out || !n_out) return TSM_ERR_INVALID_ARG;
  if (forward_index < 0 || forward_index >= (int32_t)e->timing.size())
    return fail(e, TSM_ERR_INVALID_ARG, "no timing recorded for that forward");
  const std::vector<hipEvent_t> &ev = e->timing[forward_index];
  const int n = (int)ev.size() / 2;
  *n_out = n;
  if (n > cap) return fail(e, TSM_ERR_CAPACITY, "ms_out too small");
  for (int i = 0; i < n; ++i) {
    ms_out[i] = -1.f;
    if (!ev[2 * i] || !ev[2 * i + 1]) continue;
    TSM_HIP(e, hipEventSynchronize(ev[2 * i + 1]));
    TSM_HIP(e, hipEventElapsedTime(&ms_out[i], ev[2 * i], ev[2 * i + 1]));
  }
  return TSM_OK;
}

int tsm_conv_tiles(tsm_engine *e, int32_t n_clips, int32_t *tiles_out, int32_t cap, int32_t *n_out) {
  if (!e || !tiles_out || !n_out) return TSM_ERR_INVALID_ARG;
  // launch order: stem, then per block [downsample,] conv1, conv2, conv3
  std::vector<int> order{0};
  for (const Block &b : e->blocks) {
    if (b.down >= 0) order.push_back(b.down);
    order.push_back(b.conv1);
    order.push_back(b.conv2);
    order.push_back(b.conv3);
  }
  *n_out = (int32_t)order.size();
  if ((int)order.size() > cap) return fail(e, TSM_ERR_CAPACITY, "tiles_out too small");
  auto it = e->tile_cache.find(tile_bucket(n_clips) * e->cfg.num_segments);
  for (size_t i = 0; i < order.size(); ++i) tiles_out[i] = it == e->tile_cache.end() ? 0 : it->second[order[i]];
  // Report what RUNS: a block whose conv2 carries 0x400 (conv2 + conv3 fused) or whose conv1 carries 0x800 (the whole block)
  // never reaches the conv3 + next-conv1 launch, whatever bit 0x1000 of its conv3 code says (a forced or hand-edited code; the
  // tuner itself times 0x1000 only on blocks that run conv3 as a launch of its own)  (ADVICE r4).
  if (it != e->tile_cache.end()) {
    size_t i = 1;
    for (const Block &b : e->blocks) {
      if (b.down >= 0) ++i;
      if ((tiles_out[i] & 0x800) || (tiles_out[i + 1] & 0x400)) tiles_out[i + 2] &= ~0x1000;
      i += 3;
    }
  }
  return TSM_OK;
}

float tsm_last_forward_ms(tsm_engine *e) {
  if (!e || !e->have_time) return -1.f;
  if (hipEventSynchronize(e->ev1) != hipSuccess) return -1.f;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, e->ev0, e->ev1) != hipSuccess) return -1.f;
  return ms;
}

// ---- per-op entry points (device pointers) ----------------------------------------------------

int tsm_temporal_shift(const float *x, float *y, int64_t n_frames, int32_t n_segment, int64_t hw, int32_t c,
                       int32_t fold_div, void *stream) {
  if (!x || !y || n_frames <= 0 || hw <= 0 || c <= 0 || fold_div <= 0) return TSM_ERR_INVALID_ARG;
  const int fold = c / fold_div;
  hipError_t st = tsm::launch_temporal_shift(x, y, n_frames, n_segment, hw, c, fold, static_cast<hipStream_t>(stream));
  if (st != hipSuccess) return fail(nullptr, st == hipErrorInvalidValue ? TSM_ERR_INVALID_ARG : TSM_ERR_HIP,
                                    std::string("temporal_shift: ") + hipGetErrorString(st));
  return TSM_OK;
}

int tsm_conv_bn_act(const float *x, const float *w, const float *gamma, const float *beta, const float *mean,
                    const float *var, const float *residual, float *y, int32_t n, int32_t hi, int32_t wi,
                    int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t relu, int32_t shift_segments,
                    int32_t fold_div, int32_t dtype, void *stream) {
  if (dtype != TSM_DTYPE_F32 && dtype != TSM_DTYPE_BF16X3 && dtype != TSM_DTYPE_BF16)
    return fail(nullptr, TSM_ERR_UNSUPPORTED, "bad dtype");
  const int prec = dtype == TSM_DTYPE_BF16X3 ? tsm::kPrecBf16x3 : dtype == TSM_DTYPE_BF16 ? tsm::kPrecBf16 : tsm::kPrecF32;
  const bool x3 = prec != tsm::kPrecF32;  // any non-fp32 storage format: convert at the boundary
  if (!x || !w || !gamma || !beta || !mean || !var || !y) return fail(nullptr, TSM_ERR_INVALID_ARG, "NULL pointer");
  if (k != 1 && k != 3 && k != 7) return fail(nullptr, TSM_ERR_UNSUPPORTED, "k must be 1, 3 or 7");
  if (stride != 1 && stride != 2) return fail(nullptr, TSM_ERR_UNSUPPORTED, "stride must be 1 or 2");
  const bool stem = (k == 7);
  if (stem ? (cin != 3) : (cin % 32 != 0 || (cin & (cin - 1)) != 0))
    return fail(nullptr, TSM_ERR_UNSUPPORTED, "cin must be 3 (k=7) or a power of two >= 32");
  if (prec == tsm::kPrecBf16 && !stem && cin % 64 != 0)
    return fail(nullptr, TSM_ERR_UNSUPPORTED, "TSM_DTYPE_BF16 needs cin % 64 == 0");
  if (cout % 64 != 0) return fail(nullptr, TSM_ERR_UNSUPPORTED, "cout must be a multiple of 64");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvLayer c;
  c.cin = cin; c.cout = cout; c.k = k; c.stride = stride;
  c.cp = stem ? 4 : cin;
  c.kp = (stem && x3) ? round_up(7 * 4 * 8, prec == tsm::kPrecBf16 ? 64 : 32)
                      : round_up(k * k * c.cp, prec == tsm::kPrecBf16 ? 64 : 32);
  std::vector<float> hw_((size_t)cout * cin * k * k), hg(cout), hb(cout), hm(cout), hv(cout), wp, bias;
#define TSM_HIP0(call)                                                                              \
  do {                                                                                              \
    hipError_t _st = (call);                                                                        \
    if (_st != hipSuccess) return fail(nullptr, TSM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st)); \
  } while (0)
  TSM_HIP0(hipStreamSynchronize(s));
  TSM_HIP0(hipMemcpy(hw_.data(), w, hw_.size() * sizeof(float), hipMemcpyDeviceToHost));
  TSM_HIP0(hipMemcpy(hg.data(), gamma, cout * sizeof(float), hipMemcpyDeviceToHost));
  TSM_HIP0(hipMemcpy(hb.data(), beta, cout * sizeof(float), hipMemcpyDeviceToHost));
  TSM_HIP0(hipMemcpy(hm.data(), mean, cout * sizeof(float), hipMemcpyDeviceToHost));
  TSM_HIP0(hipMemcpy(hv.data(), var, cout * sizeof(float), hipMemcpyDeviceToHost));
  if (stem && x3) {
    if (stride != 2) return fail(nullptr, TSM_ERR_UNSUPPORTED, "the bf16 formats implement the 7x7 stem for stride 2 only");
    fold_and_pack_stem_pairs(hw_.data(), hg.data(), hb.data(), hm.data(), hv.data(), cout, c.kp, &wp, &bias);
  } else {
    fold_and_pack(hw_.data(), hg.data(), hb.data(), hm.data(), hv.data(), cout, cin, k, c.cp, c.kp, &wp, &bias);
  }
  if (prec == tsm::kPrecBf16x3) to_split(&wp);
  if (prec == tsm::kPrecBf16) to_bf16(&wp);
  float *d_w = nullptr, *d_b = nullptr, *d_x4 = nullptr, *d_xs = nullptr, *d_rs = nullptr, *d_ys = nullptr;
  struct Scratch {  // frees the temporaries on every exit path (errors included)
    float **ptrs[6];
    ~Scratch() {
      for (float **q : ptrs)
        if (*q) (void)hipFree(*q);
    }
  } scratch{{&d_w, &d_b, &d_x4, &d_xs, &d_rs, &d_ys}};
  TSM_HIP0(hipMalloc(reinterpret_cast<void **>(&d_w), wp.size() * sizeof(float)));
  TSM_HIP0(hipMalloc(reinterpret_cast<void **>(&d_b), bias.size() * sizeof(float)));
  TSM_HIP0(hipMemcpy(d_w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice));
  TSM_HIP0(hipMemcpy(d_b, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  c.d_w = d_w; c.d_b = d_b;
  const float *xin = x;
  const float *rin = residual;
  float *yout = y;
  const int pad_ = k / 2;
  const size_t out_elems = (size_t)n * ((hi + 2 * pad_ - k) / stride + 1) * ((wi + 2 * pad_ - k) / stride + 1) * cout;
  if (stem) {  // NHWC3 -> NHWC4 fp32 / NHWC8 split
    TSM_HIP0(hipMalloc(reinterpret_cast<void **>(&d_x4), (size_t)n * hi * wi * 8 * sizeof(float)));
    TSM_HIP0(tsm::launch_pack_input(x, d_x4, n, hi, wi, 0, prec, s));
    xin = d_x4;
  } else if (x3) {
    TSM_HIP0(hipMalloc(reinterpret_cast<void **>(&d_xs), (size_t)n * hi * wi * cin * sizeof(float)));
    TSM_HIP0(tsm::launch_from_f32(x, d_xs, (int64_t)n * hi * wi * cin / 8, prec, s));
    xin = d_xs;
  }
  if (x3) {
    TSM_HIP0(hipMalloc(reinterpret_cast<void **>(&d_ys), out_elems * sizeof(float)));
    yout = d_ys;
    if (residual) {
      TSM_HIP0(hipMalloc(reinterpret_cast<void **>(&d_rs), out_elems * sizeof(float)));
      TSM_HIP0(tsm::launch_from_f32(residual, d_rs, (int64_t)out_elems / 8, prec, s));
      rin = d_rs;
    }
  }
  tsm::ConvParams p = make_params(c, xin, rin, yout, n, hi, wi, relu != 0, shift_segments,
                                  fold_div > 0 ? fold_div : 1, prec);
  // (a test / debug entry point that packs weights and allocates on every call: its tuning hooks are read per call)
  const char *sd_env = getenv("TSM_STEM_DIRECT");
  const int forced = tsm::conv_tile_from_name(getenv("TSM_CONV_TILE"));
  if (forced && tsm::conv_tile_valid(p, forced)) p.tile = forced;
  hipError_t st;
  if (stem && x3 && cout == 64 && !(sd_env && atoi(sd_env) == 0))
    st = tsm::launch_stem_direct(xin, c.d_w, c.d_b, yout, n, hi, wi, c.kp, relu != 0, prec, s);
  else
    st = tsm::launch_conv(p, k, s);
  if (st == hipSuccess && x3) st = tsm::launch_to_f32(d_ys, y, (int64_t)out_elems / 8, prec, s);
  hipError_t st2 = hipStreamSynchronize(s);
  if (st != hipSuccess) return fail(nullptr, st == hipErrorInvalidValue ? TSM_ERR_INVALID_ARG : TSM_ERR_HIP,
                                    std::string("launch_conv: ") + hipGetErrorString(st));
  if (st2 != hipSuccess) return fail(nullptr, TSM_ERR_HIP, std::string("conv sync: ") + hipGetErrorString(st2));
  return TSM_OK;
#undef TSM_HIP0
}

int tsm_maxpool3x3s2(const float *x, float *y, int32_t n, int32_t hi, int32_t wi, int32_t c, void *stream) {
  if (!x || !y || n <= 0 || hi <= 0 || wi <= 0) return TSM_ERR_INVALID_ARG;
  hipError_t st = tsm::launch_maxpool3x3s2(x, y, n, hi, wi, c, tsm::kPrecF32, static_cast<hipStream_t>(stream));
  if (st != hipSuccess) return fail(nullptr, st == hipErrorInvalidValue ? TSM_ERR_INVALID_ARG : TSM_ERR_HIP,
                                    std::string("maxpool: ") + hipGetErrorString(st));
  return TSM_OK;
}

int tsm_preprocess(const void *frames, int32_t pixel, int32_t n, int32_t h, int32_t w, float *out,
                   int32_t out_layout, int32_t resize, int32_t crop, int32_t scale_255, void *stream) {
  if (!frames || !out || n <= 0 || h <= 0 || w <= 0 || resize <= 0 || crop <= 0)
    return fail(nullptr, TSM_ERR_INVALID_ARG, "bad preprocess arguments");
  if (pixel != TSM_PIXEL_U8 && pixel != TSM_PIXEL_F32) return fail(nullptr, TSM_ERR_INVALID_ARG, "bad pixel type");
  if (out_layout != TSM_LAYOUT_NTHWC4 && out_layout != TSM_LAYOUT_NTCHW && out_layout != TSM_LAYOUT_NTHWC8S &&
      out_layout != TSM_LAYOUT_NTHWC8B)
    return fail(nullptr, TSM_ERR_INVALID_ARG, "out_layout must be NTHWC4, NTHWC8S, NTHWC8B or NTCHW");
  tsm::PreprocParams p{};
  p.src = frames; p.dst = out; p.n = n; p.h = h; p.w = w;
  // torchvision 0.13 Resize(int): short side -> resize, long side -> int(resize * long / short)
  if (h <= w) { p.nh = resize; p.nw = (int)((double)resize * w / h); }
  else { p.nh = (int)((double)resize * h / w); p.nw = resize; }
  if (crop > p.nh || crop > p.nw) return fail(nullptr, TSM_ERR_INVALID_ARG, "crop larger than the resized frame");
  // CenterCrop: int(round((dim - crop) / 2)) with Python's round-half-to-even
  p.top = (int)std::nearbyint((p.nh - crop) / 2.0);
  p.left = (int)std::nearbyint((p.nw - crop) / 2.0);
  p.crop = crop;
  p.src_is_u8 = pixel == TSM_PIXEL_U8;
  p.out_mode = out_layout == TSM_LAYOUT_NTCHW ? 1 : out_layout == TSM_LAYOUT_NTHWC8S ? 2
               : out_layout == TSM_LAYOUT_NTHWC8B ? 3 : 0;
  p.pre_scale = scale_255 ? 1.0f / 255.0f : 1.0f;
  hipError_t st = tsm::launch_preprocess(p, static_cast<hipStream_t>(stream));
  if (st != hipSuccess) return fail(nullptr, st == hipErrorInvalidValue ? TSM_ERR_INVALID_ARG : TSM_ERR_HIP,
                                    std::string("preprocess: ") + hipGetErrorString(st));
  return TSM_OK;
}

int tsm_gather_clips(const void *frames, int64_t n_frames, int64_t frame_bytes, int64_t first_frame, int64_t total_frames,
                     int64_t pad_frame, int64_t first_clip, int32_t n_clips, int32_t n_segment, int32_t clip_step,
                     int32_t clip_stride, void *out, void *stream) {
  if (!frames || !out) return fail(nullptr, TSM_ERR_INVALID_ARG, "gather_clips: null pointer");
  tsm::GatherParams p{};
  p.frames = frames; p.out = out; p.n_frames = n_frames; p.frame_bytes = frame_bytes; p.first_frame = first_frame;
  p.total_frames = total_frames; p.first_clip = first_clip; p.pad_frame = pad_frame; p.n_clips = n_clips;
  p.n_segment = n_segment; p.clip_step = clip_step; p.clip_stride = clip_stride;
  hipError_t st = tsm::launch_gather_clips(p, static_cast<hipStream_t>(stream));
  if (st != hipSuccess)
    return fail(nullptr, st == hipErrorInvalidValue ? TSM_ERR_INVALID_ARG : TSM_ERR_HIP,
                st == hipErrorInvalidValue ? std::string("gather_clips: a clip of the range reads outside the frame buffer, "
                                                         "the pad frame is one of the range's video frames, or frame_bytes "
                                                         "is not a multiple of 16")
                                           : std::string("gather_clips: ") + hipGetErrorString(st));
  return TSM_OK;
}

int tsm_head(const float *feat, const float *fc_w, const float *fc_b, float *logits, int32_t n_clips,
             int32_t n_segment, int32_t hw, int32_t c, int32_t num_class, void *stream) {
  if (!feat || !fc_w || !fc_b || !logits || n_clips <= 0 || n_segment <= 0 || hw <= 0 || c <= 0 || num_class <= 0)
    return TSM_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *pooled = nullptr;
  hipError_t st = hipMalloc(reinterpret_cast<void **>(&pooled), (size_t)n_clips * n_segment * c * sizeof(float));
  if (st != hipSuccess) return fail(nullptr, TSM_ERR_HIP, std::string("head scratch: ") + hipGetErrorString(st));
  st = tsm::launch_head(feat, fc_w, fc_b, pooled, logits, n_clips, n_segment, hw, c, num_class, tsm::kPrecF32, s);
  hipError_t st2 = hipStreamSynchronize(s);
  (void)hipFree(pooled);
  if (st != hipSuccess || st2 != hipSuccess)
    return fail(nullptr, TSM_ERR_HIP, std::string("head: ") + hipGetErrorString(st != hipSuccess ? st : st2));
  return TSM_OK;
}

int tsm_scores_to_states(const float *logits, int32_t n_clips, int32_t num_class, int32_t softmax, float threshold,
                         int32_t *states, float *top_score, void *stream) {
  if (!logits || !states || n_clips <= 0 || num_class <= 0)
    return fail(nullptr, TSM_ERR_INVALID_ARG, "scores_to_states: NULL pointer or non-positive size");
  hipError_t st = tsm::launch_scores_to_states(logits, n_clips, num_class, softmax != 0, threshold, states, top_score,
                                               static_cast<hipStream_t>(stream));
  if (st != hipSuccess) return fail(nullptr, TSM_ERR_HIP, std::string("scores_to_states: ") + hipGetErrorString(st));
  return TSM_OK;
}

}  // extern "C"
