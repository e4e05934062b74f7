// Probe: what the exact-fp32 MFMA sustains on this box at the conv kernel's occupancy, without memory
// traffic (A) and with one __syncthreads per 16 MFMAs (B).  Random-ish register operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool BARRIER>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
  f32x16 acc = {0};
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; b[i] = seed * 0.5f - threadIdx.x * 0.002f + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[g], acc, 0, 0, 0);
    if (BARRIER) __syncthreads();
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <bool BARRIER>
double run(int blocks, int iters) {
  float *d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<BARRIER>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<BARRIER>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f + r);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  const double flops = 5.0 * blocks * 4 /*waves*/ * (double)iters * 16 * 4096.0;
  return flops / (ms * 1e-3) / 1e12;
}
int main() {
  for (int bpc : {1, 2, 4, 5}) {
    const int blocks = 256 * bpc * 4;  // 4 rounds of bpc blocks per CU
    printf("blocks/CU %d: no barrier %.1f TF/s, barrier per 16 MFMAs %.1f TF/s\n", bpc, run<false>(blocks, 4000 / bpc),
           run<true>(blocks, 4000 / bpc));
  }
  return 0;
}
