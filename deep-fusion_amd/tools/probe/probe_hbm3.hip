// probe_hbm3.hip -- write roof vs buffer size and vs the store SHAPE of the conv epilogue:
// (a) linear 16 B/lane streams of 411 MB and 1.64 GB, plain and non-temporal;
// (b) config-5-shaped stores: units of 8 rows x 32 pixels, 512 B per pixel, row pitch
//     224 x 512 B (every wave writes 32 px x 512 B = one 16 KB row segment).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_write(v4i *dst, size_t n16, int val) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4i v = {val, val + 1, val + 2, val + 3};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}
// unit u = (image, 8-row band, 32-px column block); wave w of a 512-thread block writes row w
__global__ __launch_bounds__(512) void k_write_vgg(char *dst, int nunits, int val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  v4i v = {val, val + 1, val + 2, val + 3};
  for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
    const int n = u / (28 * 7), r = u % (28 * 7), by = r / 7, bx = r % 7;
    char *row = dst + (((size_t)n * 224 + by * 8 + wave) * 224 + bx * 32) * 512;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int pl = 8 * (e >> 2) + 4 * h + (e & 3);
      __builtin_nontemporal_store(v, (v4i *)(row + pl * 512 + l31 * 16));
    }
  }
}
int main() {
  const size_t big = (size_t)64 * 224 * 224 * 128 * 4;  // config-5 output, 1.64 GB
  char *a;
  if (hipMalloc(&a, big * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto timeit = [&](const char *name, auto f, double gb) {
    for (int i = 0; i < 2; ++i) f(i);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) f(i);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %9.2f us  %8.1f GB/s\n", name, ms * 1e3 / it, gb / (ms * 1e-3 / it));
  };
  for (size_t bytes : {(size_t)411041792, big}) {
    char nm[96];
    snprintf(nm, 96, "linear write %.0f MB plain g2048", bytes / 1e6);
    timeit(nm, [&](int i) { k_write<false><<<2048, 256>>>((v4i *)(a + (i & 1) * bytes), bytes / 16, i); }, bytes / 1e9);
    snprintf(nm, 96, "linear write %.0f MB non-temporal g2048", bytes / 1e6);
    timeit(nm, [&](int i) { k_write<true><<<2048, 256>>>((v4i *)(a + (i & 1) * bytes), bytes / 16, i); }, bytes / 1e9);
  }
  for (int grid : {256, 512, 1024})  {
    char nm[96];
    snprintf(nm, 96, "config-5-shaped 8x32 units nt, grid %d", grid);
    timeit(nm, [&](int i) { k_write_vgg<<<grid, 512>>>(a + (i & 1) * big, 64 * 28 * 7, i); }, big / 1e9);
  }
  return 0;
}
