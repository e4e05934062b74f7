// Probe: cost anatomy of the fp32 64x64 "register-resident K-step" loop (conv_igemm, RKT=1), 5 workgroups/CU.
// Each variant adds one ingredient of the real loop to a bare 16-MFMA step:
//   V0 MFMAs only            V1 + two barriers per step       V2 + 8 ds_read_b128 of fragments
//   V3 + 4 ds_write_b128     V4 + 4 buffer_load_dwordx4 (L2-resident tile, 16 KB per workgroup and step)
//   V5 = V4 with loads that stream a large buffer (HBM / MALL)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ void __launch_bounds__(256) k(const float *src, float *out, int iters, size_t span_floats) {
  __shared__ __attribute__((aligned(16))) float smem[128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < 128 * 36; i += 256) smem[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc = {0};
  f32x4 a[4], b[4], st[4];
  for (int i = 0; i < 4; ++i) { a[i] = f32x4{1.f + tid, 2.f, 3.f, 4.f}; b[i] = f32x4{0.5f, 0.25f + i, 1.f, 2.f}; st[i] = f32x4{0, 0, 0, 0}; }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)(span_floats * 4 > 0x7fffffffu ? 0x7fffffffu : span_floats * 4), 0x00020000);
  unsigned goff = (unsigned)(((size_t)blockIdx.x * 4096 + tid * 4) % span_floats) * 4u;
  for (int it = 0; it < iters; ++it) {
    if (V >= 2) {
      const float *As = smem + (wm * 32 + l31) * 36 + half * 4, *Bs = smem + (64 + wn * 32 + l31) * 36 + half * 4;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { a[kk] = *(const f32x4 *)(As + kk * 8); b[kk] = *(const f32x4 *)(Bs + kk * 8); }
    }
    if (V >= 1) __syncthreads();
    int cnt = 0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][s], b[kk][s], acc, 0, 0, 0);
        ++cnt;
        if (V >= 3 && cnt <= 4) { *(f32x4 *)(smem + ((tid >> 3) + 32 * (cnt - 1)) * 36 + (tid & 7) * 4) = st[cnt - 1]; __builtin_amdgcn_sched_barrier(0); }
        if (V >= 4 && cnt > 4 && cnt <= 8) {
          st[cnt - 5] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff, (cnt - 5) * 4096, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    if (V >= 5) goff = (unsigned)((goff + 16384u * gridDim.x) % (unsigned)(span_floats * 4));
    if (V >= 1) __syncthreads();
  }
  float r = st[0][0] + st[1][1] + st[2][2] + st[3][3];
  for (int i = 0; i < 16; ++i) r += acc[i];
  out[blockIdx.x * 256 + tid] = r;
}
template <int V>
double run(const float *src, size_t span, int blocks, int iters) {
  float *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, src, d, iters, span);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, src, d, iters, span);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return 3.0 * blocks * 4 * (double)iters * 16 * 4096.0 / (ms * 1e-3) / 1e12;
}
int main() {
  const size_t big = (size_t)1 << 30;  // 4 GiB of floats? no: 1 Gi floats = 4 GiB
  float *src; hipMalloc(&src, big);                                            // 1 GiB
  {  // random-ish non-zero data: zero operands would let the clock rise and flatter every variant
    float *h = (float *)malloc(big);
    unsigned x = 12345u;
    for (size_t i = 0; i < big / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(src, h, big, hipMemcpyHostToDevice);
    free(h);
  }
  const size_t span_small = 1 << 20, span_big = big / 4 / 4 * 4 / 1;          // 4 MiB vs 1 GiB (in floats: /4)
  const int blocks = 256 * 5 * 4, iters = 600;
  printf("V0 mfma only            %.1f TF/s\n", run<0>(src, span_small, blocks, iters));
  printf("V1 + 2 barriers         %.1f TF/s\n", run<1>(src, span_small, blocks, iters));
  printf("V2 + 8 ds_read_b128     %.1f TF/s\n", run<2>(src, span_small, blocks, iters));
  printf("V3 + 4 ds_write_b128    %.1f TF/s\n", run<3>(src, span_small, blocks, iters));
  printf("V4 + 4 buffer_load (L2) %.1f TF/s\n", run<4>(src, span_small, blocks, iters));
  printf("V5 loads stream 256 MiB %.1f TF/s (16 KB of fresh data per workgroup and step: 9x the real kernel's HBM rate)\n", run<5>(src, big / 16, blocks, iters));
  return 0;
}
