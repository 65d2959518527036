// probe_hbm2.hip -- what keeps the conv kernel's store stream below the pure-write rate?
// Variants of the conv-shaped tile store: store flavour, waves per CU, per-XCD completion times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_tiles(v4i *dst, size_t ntiles, int val, unsigned long long *tend, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4i v = {val, val + 1, val + 2, val + 3};
  for (size_t t = (size_t)blockIdx.x * WAVES + wave; t < ntiles; t += (size_t)gridDim.x * WAVES) {
    char *base = (char *)dst + t * 32768;
    const int h = lane >> 5, l31 = lane & 31;
    for (int i = 0; i < spin; ++i) asm volatile("s_nop 15");
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = 8 * (e >> 2) + 4 * h + (e & 3);
        v4i *p = (v4i *)(base + pl * 1024 + cg * 512 + l31 * 16);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
      }
  }
  if (threadIdx.x == 0) tend[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main() {
  const size_t bytes = 411041792;
  v4i *a; unsigned long long *tend;
  hipMalloc(&a, bytes * 2); hipMalloc(&tend, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch, int grid) {
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(grid);
    hipMemcpy(t.data(), tend, grid * 8, hipMemcpyDeviceToHost);
    unsigned long long mn = ~0ull; for (auto x : t) mn = x < mn ? x : mn;
    double xcd[8] = {0}; int cnt[8] = {0};
    for (int b = 0; b < grid; ++b) { xcd[b % 8] += (t[b] - mn) / 100.0; cnt[b % 8]++; }
    printf("%-34s %7.2f us %7.1f GB/s | per-XCD mean end (us after first):", name, ms * 1e3 / it, bytes / 1e9 / (ms * 1e-3 / it));
    for (int x = 0; x < 8; ++x) printf(" %.0f", xcd[x] / cnt[x]);
    printf("\n");
  };
  const size_t nt = bytes / 32768;
  run("plain 8 waves x 512 WG", [&](int i) { k_tiles<0, 8><<<512, 512>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 0); }, 512);
  run("nt    8 waves x 512 WG", [&](int i) { k_tiles<1, 8><<<512, 512>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 0); }, 512);
  run("plain 16 waves x 256 WG", [&](int i) { k_tiles<0, 16><<<256, 1024>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 0); }, 256);
  run("nt    16 waves x 256 WG", [&](int i) { k_tiles<1, 16><<<256, 1024>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 0); }, 256);
  run("plain 7 waves x 512 WG", [&](int i) { k_tiles<0, 7><<<512, 448>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 0); }, 512);
  run("nt    7 waves x 512 WG", [&](int i) { k_tiles<1, 7><<<512, 448>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 0); }, 512);
  run("nt 7x512, 4K-cycle gap/tile", [&](int i) { k_tiles<1, 7><<<512, 448>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 64); }, 512);
  run("nt 7x512, 8K-cycle gap/tile", [&](int i) { k_tiles<1, 7><<<512, 448>>>((v4i *)((char *)a + (i & 1) * bytes), nt, i, tend, 128); }, 512);
  return 0;
}
