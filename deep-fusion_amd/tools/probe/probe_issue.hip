// probe_issue.hip -- how many instructions per cycle does ONE SIMD of gfx950 issue, by instruction type and by
// the number of waves it hosts?  (Round 3: three differently structured conv kernels all ran at ~4.7 cycles per
// dynamic instruction per SIMD, counting every type -- is that a per-SIMD issue limit, and do different types
// issue side by side from different waves?)
// One workgroup per CU (128 KB of LDS keep a second one off), W waves per SIMD (block = 256 * W threads; wave w
// sits on SIMD w % 4, so waves w, w + 4, ... share a SIMD).  Wave w runs stream kind[w / 4 % nkinds]:
//   V  independent v_fma_f32             S  independent s_add_u32 / s_lshl_b32 pairs
//   L  ds_read_b32 (address 0, result unused until the end)      N  s_nop 0
//   M  dependent-chain pairs of v_mfma_i32_32x32x32_i8 (2 chains)
// Reported: cycles per instruction of each wave slot (median over CUs).
// build: hipcc -O2 --offload-arch=gfx950 probe_issue.hip -o probe_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define R8(X) X X X X X X X X
__device__ __forceinline__ void stream_v(int iters, float (&f)[8], float c) {
  for (int it = 0; it < iters; ++it) {
    R8(asm volatile("v_fma_f32 %0, %8, %0, %8\n\tv_fma_f32 %1, %8, %1, %8\n\tv_fma_f32 %2, %8, %2, %8\n\tv_fma_f32 %3, %8, %3, %8\n\t"
                    "v_fma_f32 %4, %8, %4, %8\n\tv_fma_f32 %5, %8, %5, %8\n\tv_fma_f32 %6, %8, %6, %8\n\tv_fma_f32 %7, %8, %7, %8"
                    : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c));)
  }
}
__device__ __forceinline__ void stream_s(int iters, unsigned &a0, unsigned &a1, unsigned &a2, unsigned &a3) {
  for (int it = 0; it < iters; ++it) {
    R8(asm volatile("s_add_u32 %0, %0, 1\n\ts_lshl_b32 %1, %1, 1\n\ts_add_u32 %2, %2, 3\n\ts_xor_b32 %3, %3, 5\n\t"
                    "s_add_u32 %0, %0, 7\n\ts_lshl_b32 %1, %1, 1\n\ts_add_u32 %2, %2, 9\n\ts_xor_b32 %3, %3, 11"
                    : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3)::"scc");)
  }
}
__device__ __forceinline__ void stream_l(int iters, int addr, int (&d)[8]) {
  for (int it = 0; it < iters; ++it) {
    R8(asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\t"
                    "ds_read_b32 %4, %8 offset:16\n\tds_read_b32 %5, %8 offset:20\n\tds_read_b32 %6, %8 offset:24\n\tds_read_b32 %7, %8 offset:28\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]) : "v"(addr) : "memory");)
  }
}
__device__ __forceinline__ void stream_n(int iters) {
  for (int it = 0; it < iters; ++it) {
    R8(asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");)
  }
}
__device__ __forceinline__ void stream_m(int iters, v4i a, v4i b, v16i &c0, v16i &c1) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {  // 8 MFMAs per iteration (the other streams: 64 instructions)
      __builtin_amdgcn_sched_barrier(0);
      if (m & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
    }
  }
}

// kinds: string of up to 4 letters, one per wave slot of a SIMD (slot = wave / 4); '-' = the wave exits at once
__global__ __launch_bounds__(1024) void k(unsigned *sink, unsigned long long *cyc, unsigned kinds, int iters) {
  extern __shared__ unsigned char lds[];
  const int wave = threadIdx.x >> 6, slot = wave >> 2;
  const char kind = (char)((kinds >> (8 * slot)) & 0xff);
  float f[8];
  int d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) f[i] = 1.0f + i + threadIdx.x;
  unsigned a0 = blockIdx.x, a1 = 1, a2 = 2, a3 = 3;
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x * 3, 8};
  v16i c0 = {}, c1 = {};
  if (threadIdx.x < 64) reinterpret_cast<int *>(lds)[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (kind == 'V') stream_v(iters, f, 1.0001f);
  else if (kind == 'S') stream_s(iters, a0, a1, a2, a3);
  else if (kind == 'L') stream_l(iters, 0, d);
  else if (kind == 'N') stream_n(iters);
  else if (kind == 'M') stream_m(iters, a, b, c0, c1);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned acc = a0 + a1 + a2 + a3;
  for (int i = 0; i < 8; ++i) acc += (unsigned)f[i] + d[i];
  for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = (kind == '-') ? 0 : t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int grid = prop.multiProcessorCount, iters = 300;
  unsigned *sink; unsigned long long *cyc;
  (void)hipMalloc(&sink, (size_t)grid * 1024 * 4); (void)hipMalloc(&cyc, (size_t)grid * 16 * 8);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const char *configs[] = {"V", "VV", "VVV", "VVVV", "S", "SS", "SSSS", "L", "LL", "LLLL", "N", "NNNN", "M", "MM",
                           "VS", "VL", "SL", "VSL", "VSLN", "MV", "MS", "ML", "MVS", "MVSL", "MVV", "MVVV", "MSS", "MLL"};
  for (const char *cfg : configs) {
    const int w = (int)strlen(cfg);
    unsigned kinds = 0;
    for (int i = 0; i < w; ++i) kinds |= (unsigned)(unsigned char)cfg[i] << (8 * i);
    for (int rep = 0; rep < 2; ++rep) { k<<<grid, 256 * w, 128 * 1024>>>(sink, cyc, kinds, iters); (void)hipDeviceSynchronize(); }
    std::vector<unsigned long long> c((size_t)grid * 16);
    (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    printf("{\"waves_per_simd\": \"%s\"", cfg);
    for (int s = 0; s < w; ++s) {
      std::vector<double> v;
      for (int bl = 0; bl < grid; ++bl)
        for (int q = 0; q < 4; ++q) v.push_back((double)c[(size_t)bl * 16 + 4 * s + q]);
      std::sort(v.begin(), v.end());
      const double per = cfg[s] == 'M' ? 8.0 : 64.0;
      printf(", \"%c%d_cycles_per_instr\": %.2f", cfg[s], s, v[v.size() / 2] / iters / per);
    }
    printf("}\n");
  }
  return 0;
}
