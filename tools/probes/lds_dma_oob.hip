// Probe: what does a buffer_load ... lds (LDS-DMA) write for lanes whose offset fails the range check?
// build: hipcc --offload-arch=gfx950 -O2 -o lds_dma_oob lds_dma_oob.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *x, float *y, int nbytes) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7.f;  // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, nbytes, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // odd lanes are sent out of range
  const unsigned off = (lane & 1) ? 0x80000000u : (unsigned)(threadIdx.x * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + wave * 256), 16, (int)off, 0, 0, 0);
  __syncthreads();
  *(f32x4 *)(y + threadIdx.x * 4) = *(f32x4 *)(lds + threadIdx.x * 4);
}
int main() {
  const int n = 1024;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
  float *dx, *dy;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dx, dy, n * 4);
  std::vector<float> o(n);
  hipMemcpy(o.data(), dy, n * 4, hipMemcpyDeviceToHost);
  int ok_in = 0, zero_oob = 0, stale_oob = 0, other = 0;
  for (int t = 0; t < 256; ++t)
    for (int e = 0; e < 4; ++e) {
      const float v = o[t * 4 + e];
      if (t & 1) { if (v == 0.f) ++zero_oob; else if (v == -7.f) ++stale_oob; else ++other; }
      else { if (v == h[t * 4 + e]) ++ok_in; else ++other; }
    }
  printf("in-range correct %d/512, out-of-range: zero %d, untouched(sentinel) %d, other %d\n", ok_in, zero_oob, stale_oob, other);
  return 0;
}
