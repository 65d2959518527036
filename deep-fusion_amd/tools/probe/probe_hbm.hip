// probe_hbm.hip -- calibrates the HBM roof on this box: write-only and copy streams
// with 16 B/lane accesses (the store shape of the conv epilogue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_write(v4i *dst, size_t n16, int val) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4i v = {val, val + 1, val + 2, val + 3};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(v4i *dst, const v4i *src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
// strided store shape of the conv kernel: each wave writes 2 x 512 B segments 4 KB apart
__global__ __launch_bounds__(512) void k_write_tiles(v4i *dst, size_t ntiles, int val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4i v = {val, val + 1, val + 2, val + 3};
  for (size_t t = (size_t)blockIdx.x * 8 + wave; t < ntiles; t += (size_t)gridDim.x * 8) {
    char *base = (char *)dst + t * 32768;  // 32 px x 1 KB
    const int h = lane >> 5, l31 = lane & 31;
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = 8 * (e >> 2) + 4 * h + (e & 3);
        *(v4i *)(base + pl * 1024 + cg * 512 + l31 * 16) = v;
      }
  }
}
int main() {
  const size_t bytes = 411041792;  // config-3 output
  v4i *a, *b;
  hipMalloc(&a, bytes * 2); hipMalloc(&b, bytes * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *name, auto f, double gb) {
    for (int i = 0; i < 3; ++i) f(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) f(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.2f us  %8.1f GB/s\n", name, ms * 1e3 / it, gb / (ms * 1e-3 / it));
  };
  for (int grid : {512, 1024, 2048, 4096}) {
    char nm[64]; snprintf(nm, 64, "write 411MB grid %d", grid);
    timeit(nm, [&](int i) { k_write<<<grid, 256>>>((v4i *)((char *)a + (i & 1) * bytes), bytes / 16, i); }, bytes / 1e9);
  }
  timeit("copy 411MB->411MB grid 2048", [&](int i) { k_copy<<<2048, 256>>>(b, a, bytes / 16); }, 2 * bytes / 1e9);
  timeit("conv-shaped tile stores g512", [&](int i) { k_write_tiles<<<512, 512>>>((v4i *)((char *)a + (i & 1) * bytes), bytes / 32768, i); }, bytes / 1e9);
  timeit("conv-shaped tile stores g1024", [&](int i) { k_write_tiles<<<1024, 512>>>((v4i *)((char *)a + (i & 1) * bytes), bytes / 32768, i); }, bytes / 1e9);
  return 0;
}
