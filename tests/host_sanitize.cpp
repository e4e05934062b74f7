// ASAN + UBSAN harness for the pure-host pieces of the engine (workoutdetector_amd/csrc/tsm_host_util.h): built by
// tests/test_host_sanitizers.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all  and run on the CPU.
// (GPU AddressSanitizer is not available on this pool; the device code is covered by the parity tests instead.)
//
//   1. fuzz loop over malformed TSM_TUNE_CACHE lines through parse_tune_line (the hand-written parser that reads
//      a user-supplied file): truncated lines, huge numbers, stray bytes, missing separators, wrong code counts.
//   2. fold_and_pack / fold_and_pack_stem_pairs / to_split / to_bf16 on ragged sizes, with the packed-buffer
//      invariants checked (every weight lands inside [cout][kp], padding stays zero, split hi+lo == value to 2^-16).
#include <cstdio>
#include <random>

#include "../workoutdetector_amd/csrc/tsm_host_util.h"

using namespace tsm_host;

static int failures = 0;
#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

static void fuzz_tune_lines(unsigned seed, int rounds) {
  std::mt19937 rng(seed);
  const std::string want = "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|";
  const int kNumTiles = 7;
  // a well-formed line first
  {
    std::vector<int> codes(5, -1);
    const std::string line = want + "3,259,4,1,5\n";
    EXPECT(parse_tune_line(line.c_str(), want, kNumTiles, &codes));
    EXPECT(codes[0] == 3 && codes[1] == 259 && codes[2] == 4 && codes[3] == 1 && codes[4] == 5);
  }
  // the fusion bits survive a round trip: 0x400 (conv2 + conv3 as one launch) and 0x800 (the whole block as one launch;
  // a parser that dropped such a line would make every rank of a multi-GPU job tune for itself again)
  {
    std::vector<int> codes(5, -1);
    const std::string line = want + "2054,1030,6,3075,4102\n";       // (4102 = 6 | 0x1000: conv3 also runs the next block's conv1)
    EXPECT(parse_tune_line(line.c_str(), want, kNumTiles, &codes));
    EXPECT(codes[0] == (6 | 0x800) && codes[1] == (6 | 0x400) && codes[2] == 6 && codes[3] == (3 | 0x400 | 0x800) && codes[4] == (6 | 0x1000));
    const std::string line2 = want + "515,8198,3,3,3\n";             // (515 = 3 | 0x200: tail split; 8198 = 6 | 0x2000: conv1 also runs the stride-2 conv2)
    EXPECT(parse_tune_line(line2.c_str(), want, kNumTiles, &codes));
    EXPECT(codes[0] == (3 | 0x200) && codes[1] == (6 | 0x2000));
  }
  const char *bad[] = {"", "\n", "|", "abi3", "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|",
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3",            // too few
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3,3,3",        // too many
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3,9",          // tile out of range
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3,35",         // reserved bits set
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3,16387",      // a bit above the fusion bits
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3,-1",
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,3,3,3,99999999999999999999999999",
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3,,3,3,3",
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|3 3 3 3 3",
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|256|9999,abc",
                       "abi3 gfx950 T8 224x224 dtype0 shift8 fuse1|128|3,3,3,3,3"};        // another bucket
  for (const char *b : bad) {
    std::vector<int> codes(5, -7);
    EXPECT(!parse_tune_line(b, want, kNumTiles, &codes));
    for (int c : codes) EXPECT(c == -7);   // untouched on failure
  }
  // random mutations of a good line and random byte strings
  const std::string good = want + "3,259,4,1,5\n";
  for (int r = 0; r < rounds; ++r) {
    std::string line = (rng() & 1) ? good : std::string();
    const int edits = 1 + (int)(rng() % 8);
    for (int k = 0; k < edits; ++k) {
      const int op = (int)(rng() % 4);
      const size_t pos = line.empty() ? 0 : rng() % (line.size() + 1);
      if (op == 0 || line.empty()) line.insert(pos, 1, (char)(1 + rng() % 255));
      else if (op == 1) line.erase(pos < line.size() ? pos : line.size() - 1, 1 + rng() % 4);
      else if (op == 2) line[pos < line.size() ? pos : line.size() - 1] = (char)(1 + rng() % 255);
      else line.insert(pos, "0123456789,|-\n"[rng() % 14] == 0 ? "" : std::string(1 + rng() % 30, "0123456789,|-\n"[rng() % 14]));
    }
    if (line.size() > 4000) line.resize(4000);
    std::vector<int> codes(5, -7);
    const bool ok = parse_tune_line(line.c_str(), want, kNumTiles, &codes);
    for (int c : codes) EXPECT(ok ? (c >= 0 && (c & ~0x3F0F) == 0 && (c & 15) < kNumTiles) : c == -7);
  }
}

static void check_packing(unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> uni(-2.f, 2.f);
  const int cases[][4] = {{64, 64, 1, 64}, {64, 64, 3, 576}, {128, 256, 1, 256}, {64, 3, 7, 224}, {96, 32, 3, 320}};
  for (const auto &c : cases) {
    const int cout = c[0], cin = c[1], k = c[2], kp = c[3], cp = k == 7 ? 4 : cin;
    std::vector<float> w((size_t)cout * cin * k * k), g(cout), b(cout), m(cout), v(cout), wp, bias;
    for (float &x : w) x = uni(rng);
    for (int o = 0; o < cout; ++o) { g[o] = 1.f + 0.25f * uni(rng); b[o] = uni(rng); m[o] = uni(rng); v[o] = 0.5f + std::fabs(uni(rng)); }
    fold_and_pack(w.data(), g.data(), b.data(), m.data(), v.data(), cout, cin, k, cp, kp, &wp, &bias);
    EXPECT(wp.size() == (size_t)cout * kp && (int)bias.size() == cout);
    const float s0 = g[0] / std::sqrt(v[0] + kBnEps);
    EXPECT(wp[0] == w[0] * s0);                                  // (o=0, ky=0, kx=0, c=0)
    for (int kk = k * k * cp; kk < kp; ++kk) EXPECT(wp[kk] == 0.f);  // K padding of row 0 stays zero
    std::vector<float> split = wp, half = wp;
    to_split(&split);
    EXPECT(split.size() == wp.size());
    const uint16_t *sp = reinterpret_cast<const uint16_t *>(split.data());
    for (size_t i = 0; i + 8 <= wp.size(); i += 8)
      for (int e = 0; e < 8; ++e) {
        const float back = bf2f(sp[2 * i + e]) + bf2f(sp[2 * i + 8 + e]);
        EXPECT(std::fabs(back - wp[i + e]) <= std::ldexp(std::fabs(wp[i + e]), -15) + 1e-30f);
      }
    to_bf16(&half);
    EXPECT(half.size() == (wp.size() + 1) / 2);
    if (k == 7) {
      std::vector<float> wq, bq;
      fold_and_pack_stem_pairs(w.data(), g.data(), b.data(), m.data(), v.data(), cout, 224, &wq, &bq);
      EXPECT(wq.size() == (size_t)cout * 224);
      EXPECT(wq[0] == 0.f);                                      // (ky 0, pair 0, pixel 0) = kx -1: zero weight
      EXPECT(wq[4] == w[0] * s0);                                // (ky 0, pair 0, pixel 1, c 0) = kx 0
      for (size_t i = 3; i < wq.size(); i += 4) EXPECT(wq[i] == 0.f);   // channel 3 is padding everywhere
    }
  }
  // conv3 weights in MFMA-fragment order (fused conv2 + conv3 kernel): a permutation of the packed matrix -- every
  // weight lands exactly once, at the slot the kernel's lane / k-group arithmetic reads it from
  for (int cmid : {64, 128}) {
    std::vector<float> w3((size_t)4 * cmid * cmid), frag, fsplit;
    for (size_t i = 0; i < w3.size(); ++i) w3[i] = (float)i + 0.25f;      // distinct, exactly representable
    pack_w3_fragments(w3.data(), cmid, &frag);
    EXPECT(frag.size() == w3.size());
    std::vector<char> seen(w3.size(), 0);
    const int wgn = cmid / 32, nkk = cmid / 8;
    for (int j = 0; j < 4; ++j)
      for (int wn = 0; wn < wgn; ++wn)
        for (int kk = 0; kk < nkk; ++kk)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const size_t src = (size_t)(j * cmid + wn * 32 + (lane & 31)) * cmid + 8 * kk + 4 * (lane >> 5) + e;
              EXPECT(frag[((((size_t)j * wgn + wn) * nkk + kk) * 64 + lane) * 4 + e] == w3[src]);
              EXPECT(!seen[src]);
              seen[src] = 1;
            }
    for (char c : seen) EXPECT(c);
    // split-bf16 form: hi / lo halves of the 8 channels of a k16 group, identical to what to_split() stores
    for (float &x : w3) x = uni(rng);
    pack_w3_fragments_split(w3.data(), cmid, &fsplit);
    EXPECT(fsplit.size() == w3.size());
    std::vector<float> ref = w3;
    to_split(&ref);                                            // groups of 8 consecutive k of one row: [hi x8 | lo x8]
    const uint16_t *fs = reinterpret_cast<const uint16_t *>(fsplit.data()), *rs = reinterpret_cast<const uint16_t *>(ref.data());
    const int nkq = cmid / 16;
    for (int j = 0; j < 4; ++j)
      for (int wn = 0; wn < wgn; ++wn)
        for (int kq = 0; kq < nkq; ++kq)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const size_t n = (size_t)j * cmid + wn * 32 + (lane & 31), k0 = 16 * kq + 8 * (lane >> 5);
              const size_t grp = (n * cmid + k0) / 8;                         // 8-element group index in to_split's layout
              const size_t base = ((((size_t)j * wgn + wn) * (2 * nkq) + 2 * kq) * 64 + lane) * 8;
              EXPECT(fs[base + e] == rs[grp * 16 + e]);
              EXPECT(fs[base + 64 * 8 + e] == rs[grp * 16 + 8 + e]);
            }
  }
  // NaN / inf survive the bf16 conversion as NaN / inf (never as a finite number)
  EXPECT(std::isnan(bf2f(f2bf(std::nanf("")))));
  EXPECT(std::isinf(bf2f(f2bf(INFINITY))));
  EXPECT(segment_len(4608, 0) > 0 && segment_len(4608, 1) == 0 && segment_len(512, 0) == 0);
  EXPECT(tile_bucket(1) == 1 && tile_bucket(5) == 8 && tile_bucket(32) == 32 && tile_bucket(33) == 64);
}

// tail_split_point: where a segmented 64x64 launch is cut into whole-K tiles and (tile, K segment) pieces.
static void check_tail_split() {
  using tsm_host::tail_split_point;
  const size_t big = (size_t)16 << 20;
  // headline shape, 256 CUs (rounds of 1 280 tiles): layer3 3x3 (M = 50 176, Cout 256: 3 136 tiles = 2.45 rounds), layer4 3x3
  // (M = 12 544, Cout 512: 1 568 = 1.225 rounds)
  EXPECT(tail_split_point(50176, 256, 4, 256, big) == 2560);
  EXPECT(tail_split_point(12544, 512, 9, 256, big) == 1280);
  // the result is a multiple of ntn, inside (0, tiles), and the scratch holds the tail's segment sums -- for every shape
  std::mt19937 rng(7);
  for (int r = 0; r < 20000; ++r) {
    const long m = 1 + (long)(rng() % 400000);
    const int cout = 64 * (1 + (int)(rng() % 32)), nseg = 1 + (int)(rng() % 12), ncu = 1 + (int)(rng() % 320);
    const size_t scratch = (size_t)(rng() % (32u << 20));
    const long from = tail_split_point(m, cout, nseg, ncu, scratch);
    const long ntn = cout / 64, tiles = (m + 63) / 64 * ntn, slots = 5L * ncu;
    if (from == 0) continue;
    EXPECT(nseg >= 2 && from > 0 && from < tiles && from % ntn == 0 && from % 1 == 0);
    EXPECT(from <= tiles / slots * slots && tiles / slots >= 1);                       // only whole rounds stay whole-K
    EXPECT((tiles - tiles / slots * slots) * 100 <= slots * 85);                      // a nearly full last round is left alone
    EXPECT((size_t)nseg * (size_t)(m - from / ntn * 64) * (size_t)cout <= scratch);   // the segment sums of the tail rows fit
  }
  EXPECT(tail_split_point(50176, 256, 4, 256, 1000) == 0);          // scratch too small
  EXPECT(tail_split_point(5488, 512, 9, 256, big) == 0);            // 688 tiles: no whole round
  EXPECT(tail_split_point(81920, 256, 4, 256, big) == 0);           // 5 120 tiles: exactly four rounds
  EXPECT(tail_split_point(50176, 256, 1, 256, big) == 0);           // one segment: nothing to split
  EXPECT(tail_split_point(50176, 96, 4, 256, big) == 0 && tail_split_point(0, 256, 4, 256, big) == 0 && tail_split_point(50176, 256, 4, 0, big) == 0);
}

int main(int argc, char **argv) {
  check_tail_split();
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
  fuzz_tune_lines(1234, rounds);
  check_packing(99);
  if (failures) {
    std::fprintf(stderr, "%d failures\n", failures);
    return 1;
  }
  std::printf("host sanitize ok (%d fuzz rounds)\n", rounds);
  return 0;
}
