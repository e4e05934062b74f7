// Device-side helpers shared by the kernel families of libtsm_hip.so (one translation unit per family: an edit to one
// family recompiles one object) and the internal launch functions they call across files.  gfx950 only.
//
//   tsm_igemm.hip      conv_igemm (every precision, every tile), launch_conv's dispatch, the split-K reduction
//   tsm_bf16_256.hip   conv_bf16_256[p]_kernel: the 256 x 256 LDS-DMA tile, one-shot and persistent
//   tsm_ws.hip         weight-stationary bf16 kernels: conv3x3_ws[128], conv1x1_ws[n]
//   tsm_bneck.hip      bneck_ws_kernel: a whole layer1 Bottleneck per launch (bf16)
//   tsm_conv31.hip     conv31_fused_kernel: conv3 + residual of block b and shift + conv1 of block b + 1 per launch (bf16)
//   tsm_front.hip      front_s2_kernel: shift + conv1 + stride-2 conv2 of layer2.0 per launch (bf16)
//   tsm_fused23.hip    conv23_fused_kernel: conv2 + conv3 + residual per launch (fp32 / split-bf16)
//   tsm_stem.hip       stem_direct / stem_pool[_f32]: the 7x7 stem with the max-pool fused behind it
//   tsm_ops.hip        pack / convert / preprocess / gather_clips / maxpool / shift / head / scores_to_states, device_info()
#pragma once
#include "tsm_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace tsm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS row stride in floats: 32 data + 4 pad.  With ds_read_b128 (16-lane groups, 64 banks) the 16
// rows of a group land on 16 distinct 4-bank slots (row*36 mod 64 is a permutation of multiples
// of 4), so fragment reads are conflict-free; ds_write_b128 of 8 consecutive lanes covers one row.
constexpr int kLds = 36;
constexpr int kBK = 32;

constexpr unsigned kInvalid = 0x80000000u;  // >= num_records of every descriptor below

// XCD-chunked tile order.  Workgroup b of a launch runs on XCD b & 7 (round-robin dispatch), and each XCD has an L2 of its
// own: virtual index v = b + k * gridDim.x (gridDim.x a multiple of 8, or a single pass) is mapped to a tile such that every
// XCD walks ONE contiguous eighth of the n tiles -- tiles that share halo lines, weights or neighbouring frames are then
// fetched once per XCD instead of once per workgroup.  A bijection on [0, n) for any grid.
__device__ __forceinline__ long xcd_chunked(long v, long n) {
  const long q8 = n >> 3, r8 = n & 7, x = v & 7;
  return (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (v >> 3);
}

// Cache policy of the activation stores of the bf16 kernel families (the aux operand of raw_buffer_store: 0 = plain,
// 2 = nt, 16 = sc1 = write-through that drops the line from the XCD's L2, MI355X_MICROARCH.md "stores of each flavour").
#ifndef TSM_OUT_AUX
#define TSM_OUT_AUX 0
#endif
#ifndef TSM_AUX_256
#define TSM_AUX_256 TSM_OUT_AUX
#endif
#ifndef TSM_AUX_C31
#define TSM_AUX_C31 TSM_OUT_AUX
#endif
#ifndef TSM_AUX_WS
#define TSM_AUX_WS TSM_OUT_AUX
#endif
#ifndef TSM_AUX_BNECK
#define TSM_AUX_BNECK TSM_OUT_AUX
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// split-bf16 helpers.  A 32-byte group is [hi x8 | lo x8]; word w of a half holds elements 2w (low 16
// bits) and 2w+1 (high 16 bits).
__device__ __forceinline__ float split_elem(u32x4 half8, int e) {
  const unsigned w = half8[e >> 1];
  return __builtin_bit_cast(float, (e & 1) ? (w & 0xFFFF0000u) : (w << 16));
}
__device__ __forceinline__ unsigned pack_bf16(float x0, float x1) {  // element 0 in the low half
  const bf16x2 h = {(__bf16)x0, (__bf16)x1};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned *hi, unsigned *lo) {
  const bf16x2 h = {(__bf16)x0, (__bf16)x1};  // v_cvt_pk_bf16_f32, round to nearest even
  const unsigned hw = __builtin_bit_cast(unsigned, h);
  const float r0 = x0 - __builtin_bit_cast(float, hw << 16);
  const float r1 = x1 - __builtin_bit_cast(float, hw & 0xFFFF0000u);
  const bf16x2 l = {(__bf16)r0, (__bf16)r1};
  *hi = hw;
  *lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}
// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N - 1>) -- unrolled in the AST, for bodies too large for
// `#pragma unroll` to honour (its size threshold silently leaves a loop, and the register arrays go to scratch).
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// `s_waitcnt vmcnt(n)` for a compile-time-foldable n (the instruction takes an immediate).
__device__ __forceinline__ void wait_vmcnt(int n) {
#define TSM_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    TSM_VMCNT_CASE(0) TSM_VMCNT_CASE(4) TSM_VMCNT_CASE(8) TSM_VMCNT_CASE(12) TSM_VMCNT_CASE(16) TSM_VMCNT_CASE(20)
    TSM_VMCNT_CASE(24) TSM_VMCNT_CASE(28) TSM_VMCNT_CASE(32) TSM_VMCNT_CASE(36) TSM_VMCNT_CASE(40) TSM_VMCNT_CASE(44)
    TSM_VMCNT_CASE(48) TSM_VMCNT_CASE(52) TSM_VMCNT_CASE(56) TSM_VMCNT_CASE(60)
    TSM_VMCNT_CASE(3) TSM_VMCNT_CASE(7) TSM_VMCNT_CASE(11) TSM_VMCNT_CASE(15) TSM_VMCNT_CASE(19) TSM_VMCNT_CASE(23)
    TSM_VMCNT_CASE(27) TSM_VMCNT_CASE(31) TSM_VMCNT_CASE(35) TSM_VMCNT_CASE(39) TSM_VMCNT_CASE(43) TSM_VMCNT_CASE(47)
    TSM_VMCNT_CASE(51) TSM_VMCNT_CASE(55) TSM_VMCNT_CASE(59) TSM_VMCNT_CASE(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef TSM_VMCNT_CASE
}
// Every kernel launch of the library goes through this macro: the launch itself, plus one line in the calling thread's
// launch trace when a test has switched it on (tsm_trace_launches; note_launch is a thread-local pointer test otherwise).
// The trace names the kernel as the launch site spells it and, for a launch inside a template, the enclosing
// function's template arguments -- so a parity test can assert that the kernel under test is the one that ran.
#define TSM_KLAUNCH(kern, ...)                          \
  do {                                                  \
    ::tsm::note_launch(#kern, __PRETTY_FUNCTION__);     \
    hipLaunchKernelGGL(kern, __VA_ARGS__);              \
  } while (0)

inline unsigned grid_for(int64_t total, int cap) {
  const int64_t blocks = (total + 255) / 256;
  return (unsigned)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

// Per DEVICE, once: the CU count that sizes the persistent grids and the > 64 KB dynamic-LDS opt-in of every kernel
// that needs one.  tsm_hip.h lets engines on several devices live in one process, so neither may be cached from
// whichever device happened to launch first.  device_info() (tsm_ops.hip) runs every family's opt-in function once per
// device: each returns the first error of its hipFuncSetAttribute calls.
struct DeviceInfo {
  int n_cu = 256;
  hipError_t status = hipSuccess;
};
const DeviceInfo &device_info();
hipError_t lds_opt_in(const void *kernel, size_t bytes);   // hipFuncSetAttribute(MaxDynamicSharedMemorySize)
hipError_t opt_in_bf16_256();
hipError_t opt_in_ws();
hipError_t opt_in_bneck();
hipError_t opt_in_conv31();
hipError_t opt_in_front();

// launch functions one family's dispatch calls in another family's file
hipError_t launch_conv_bf16_256(ConvParams p, int ks, hipStream_t s);
hipError_t launch_conv_bf16_256p(ConvParams p, int ks, hipStream_t s);
hipError_t launch_conv3x3_ws(ConvParams p, hipStream_t s);
hipError_t launch_conv1x1_ws(const ConvParams &p, hipStream_t s);
hipError_t launch_conv1x1_wsn(const ConvParams &p, hipStream_t s);
hipError_t launch_conv23_ws(const Fused23Params &p, hipStream_t s);

}  // namespace tsm
