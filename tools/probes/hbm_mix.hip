// Probe: what this chip sustains for pure writes, pure reads and read : write mixes of 16-byte accesses from persistent workgroups
// (one 256-thread workgroup per CU x WG_PER_CU), i.e. the memory-side ceilings the HBM-bound launches of DESIGN.md are priced against.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/hbm_mix tools/probes/hbm_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// every workgroup streams chunks of 64 KB (256 threads x 16 iterations x 16 B): reads `rd` chunks for every `wr` chunks it writes
template <int RD, int WR>
__global__ void __launch_bounds__(256) mix(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, long chunks_r, long chunks_w, unsigned *sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const long nwg = gridDim.x;
  long cr = blockIdx.x, cw = blockIdx.x;
  while ((RD && cr < chunks_r) || (WR && cw < chunks_w)) {
#pragma unroll
    for (int r = 0; r < RD; ++r) {
      if (cr < chunks_r) {
        const u32x4 *p = src + cr * 4096 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const u32x4 v = __builtin_nontemporal_load(p + i * 256); acc ^= v; }
      }
      cr += nwg;
    }
#pragma unroll
    for (int w = 0; w < WR; ++w) {
      if (cw < chunks_w) {
        u32x4 *p = dst + cw * 4096 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i * 256] = acc + (unsigned)i;
      }
      cw += nwg;
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

template <int RD, int WR>
static void run(const char *name, const u32x4 *src, u32x4 *dst, long bytes_r, long bytes_w, int grid, unsigned *sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix<RD, WR>), dim3(grid), dim3(256), 0, 0, src, dst, RD ? bytes_r / 65536 : 0, WR ? bytes_w / 65536 : 0, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double gb = ((RD ? bytes_r : 0) + (WR ? bytes_w : 0)) / 1e9;
  printf("%-28s grid %5d: %7.3f ms  read %.2f GB  write %.2f GB  -> %.2f TB/s total (%.2f read, %.2f write)\n", name, grid, best,
         RD ? bytes_r / 1e9 : 0.0, WR ? bytes_w / 1e9 : 0.0, gb / best, (RD ? bytes_r / 1e9 : 0.0) / best, (WR ? bytes_w / 1e9 : 0.0) / best);
}

int main(int argc, char **argv) {
  const long GB2 = 2L << 30;
  u32x4 *src, *dst; unsigned *sink;
  if (hipMalloc(&src, 2 * GB2) != hipSuccess || hipMalloc(&dst, 2 * GB2) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, 2 * GB2); hipMemset(dst, 0, 2 * GB2);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  for (int per : {1, 2, 4, 8}) {
    const int grid = ncu * per;
    printf("== %d workgroup(s) of 256 threads per CU\n", per);
    run<0, 1>("write only", src, dst, 0, 2 * GB2, grid, sink);
    run<1, 0>("read only", src, dst, 2 * GB2, 0, grid, sink);
    run<1, 1>("copy (1 : 1)", src, dst, GB2, GB2, grid, sink);
    run<1, 4>("read : write 1 : 4", src, dst, GB2 / 2, 2 * GB2, grid, sink);
    run<4, 1>("read : write 4 : 1", src, dst, 2 * GB2, GB2 / 2, grid, sink);
  }
  return 0;
}
