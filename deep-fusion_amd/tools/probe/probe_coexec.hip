// probe_coexec.hip -- can a SIMD of gfx950 run one wave's MFMAs and ANOTHER wave's VALU instructions at the
// same time?  (The design question behind conv_mfma_roles.cuh: round 2 measured 17 % MFMA/VALU co-execution
// with four homogeneous waves per SIMD.)
// One 512-thread workgroup per CU (8 waves: waves w and w + 4 share a SIMD; 128 KB of LDS keep a second
// workgroup off the CU).  Waves 0-3 = role M, waves 4-7 = role V.  Streams, all on int8 32x32x32 MFMAs in two
// dependent chains (like the conv0 loop) and v_pk_mul_f32 / v_cvt_pk_u8_f32 on 16 independent registers:
//   m_only      M: MFMA, MFMA, ...              V: absent
//   v_only      M: absent                       V: VALU, VALU, ...
//   both        M: MFMA stream                  V: VALU stream          <- cross-wave co-execution
//   mixK        M: (MFMA + K VALU) repeated     V: absent               <- same-wave co-execution, K = 2, 4, 6, 8
//   mixK_x2     M and V: both the mixK stream                          <- two interleaved streams per SIMD
//   gapK_both   M: (MFMA + K VALU) repeated     V: VALU stream          <- does V fill what M leaves?
// Reported per role: cycles (s_memtime) per MFMA / per VALU instruction of the wave, median over all CUs.
// build: hipcc -O2 --offload-arch=gfx950 probe_coexec.hip -o probe_coexec
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// filler kinds F: 0 = v_pk_mul_f32 + v_cvt_pk_u8_f32 (the round-2 epilogue's mix), 1 v_pk_mul_f32, 2 v_mul_f32,
// 3 v_fma_f32, 4 v_cvt_pk_u8_f32, 5 v_add_u32, 6 v_pk_fma_f32, 7 v_pk_add_f32, 8 v_cvt_f32_i32, 9 v_xor_b32
#define VALU2(i)                                                                                              \
  if (F == 0) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[(i) & 7]) : "v"(c2));                          \
                asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[(i) & 7]) : "v"(f[(i) & 7])); }          \
  if (F == 1) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[(i) & 7]) : "v"(c2));                          \
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[((i) + 4) & 7]) : "v"(c2)); }                  \
  if (F == 2) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[(i) & 7]) : "v"(c2[0]));                          \
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[((i) + 4) & 7]) : "v"(c2[0])); }                  \
  if (F == 3) { asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(f[(i) & 7]) : "v"(c2[0]));                      \
                asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(f[((i) + 4) & 7]) : "v"(c2[0])); }              \
  if (F == 4) { asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[(i) & 7]) : "v"(f[(i) & 7]));            \
                asm volatile("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(u[((i) + 4) & 7]) : "v"(f[(i) & 7])); }    \
  if (F == 5) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[(i) & 7]) : "v"(u[((i) + 1) & 7]));               \
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[((i) + 4) & 7]) : "v"(u[((i) + 5) & 7])); }       \
  if (F == 6) { asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(p[(i) & 7]) : "v"(c2));                      \
                asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(p[((i) + 4) & 7]) : "v"(c2)); }              \
  if (F == 7) { asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[(i) & 7]) : "v"(c2));                          \
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[((i) + 4) & 7]) : "v"(c2)); }                  \
  if (F == 8) { asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[(i) & 7]) : "v"(u[(i) & 7]));                     \
                asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[((i) + 4) & 7]) : "v"(u[((i) + 4) & 7])); }       \
  if (F == 9) { asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(u[(i) & 7]));                               \
                asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(u[((i) + 4) & 7])); }

template <int K, int F>  // K VALU instructions of kind F after each MFMA (K even)
__device__ __forceinline__ void mix_stream(int iters, v4i a, v4i b, v16i &c0, v16i &c1, v2f (&p)[8], unsigned (&u)[8],
                                           float (&f)[8], v2f c2) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      if (m & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K / 2; ++k) { VALU2(m * (K / 2) + k) }
    }
  }
}

template <int F>
__device__ __forceinline__ void valu_stream(int iters, v2f (&p)[8], unsigned (&u)[8], float (&f)[8], v2f c2) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) { VALU2(k) }  // 32 VALU instructions
  }
}

// mode: 0 m_only, 1 v_only, 2 both, 3 mixK (M only), 4 mixK both roles, 5 M = mixK and V = VALU stream
template <int K, int F>
__global__ __launch_bounds__(512) void k(unsigned *sink, unsigned long long *cyc, int mode, int iters) {
  extern __shared__ unsigned char lds[];
  const int wave = threadIdx.x >> 6;
  const bool roleM = wave < 4;
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x * 3, 8};
  v16i c0 = {}, c1 = {};
  v2f p[8];
  unsigned u[8];
  float f[8];
  for (int i = 0; i < 8; ++i) { p[i] = v2f{1.0f + i, 2.0f + threadIdx.x}; u[i] = i; f[i] = 3.5f * i + threadIdx.x; }
  const v2f c2 = {1.0001f, 0.9999f};
  if (threadIdx.x == 0) lds[0] = 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  bool ran = true;
  if (mode == 0) { if (roleM) mix_stream<0, F>(iters, a, b, c0, c1, p, u, f, c2); else ran = false; }
  else if (mode == 1) { if (!roleM) valu_stream<F>(iters, p, u, f, c2); else ran = false; }
  else if (mode == 2) { if (roleM) mix_stream<0, F>(iters, a, b, c0, c1, p, u, f, c2); else valu_stream<F>(iters, p, u, f, c2); }
  else if (mode == 3) { if (roleM) mix_stream<K, F>(iters, a, b, c0, c1, p, u, f, c2); else ran = false; }
  else if (mode == 4) { mix_stream<K, F>(iters, a, b, c0, c1, p, u, f, c2); }
  else { if (roleM) mix_stream<K, F>(iters, a, b, c0, c1, p, u, f, c2); else valu_stream<F>(iters, p, u, f, c2); }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
  for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
  for (int i = 0; i < 8; ++i) acc += (unsigned)p[i][0] + (unsigned)p[i][1] + u[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = ran ? t1 - t0 : 0;
}

static double median(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

static const char *kFillers[] = {"pk_mul+cvt_pk_u8", "v_pk_mul_f32", "v_mul_f32", "v_fma_f32", "v_cvt_pk_u8_f32", "v_add_u32",
                                 "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_f32_i32", "v_xor_b32 literal"};
template <int K, int F>
static void run(const char *name, int mode, unsigned *sink, unsigned long long *cyc, int grid) {
  const int iters = 400;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k<K, F>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int rep = 0; rep < 2; ++rep) { k<K, F><<<grid, 512, 128 * 1024>>>(sink, cyc, mode, iters); hipDeviceSynchronize(); }
  std::vector<unsigned long long> c(grid * 8);
  hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> m, v;
  for (int bl = 0; bl < grid; ++bl)
    for (int w = 0; w < 8; ++w) if (c[bl * 8 + w]) (w < 4 ? m : v).push_back((double)c[bl * 8 + w]);
  // per wave: M streams issue 8 MFMAs + 8 * K VALU per iteration; V streams 32 VALU per iteration (or the M stream in mode 4)
  const double mm = median(m) / iters, vv = median(v) / iters;
  const bool v_is_mix = mode == 4;
  printf("{\"stream\": \"%s\", \"filler\": \"%s\", \"K\": %d, \"M_cycles_per_mfma\": %.1f, \"M_valu_per_mfma\": %d, \"V_cycles_per_valu\": %.2f, \"V_cycles_per_mfma\": %.1f}\n",
         name, kFillers[F], K, m.empty() ? 0.0 : mm / 8, K, (v.empty() || v_is_mix) ? 0.0 : vv / 32, v_is_mix ? vv / 8 : 0.0);
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int grid = prop.multiProcessorCount;
  unsigned *sink; unsigned long long *cyc;
  hipMalloc(&sink, (size_t)grid * 512 * 4); hipMalloc(&cyc, (size_t)grid * 8 * 8);
  run<0, 0>("m_only", 0, sink, cyc, grid);
#define SWEEP(F)                                      \
  run<0, F>("v_only", 1, sink, cyc, grid);            \
  run<0, F>("both", 2, sink, cyc, grid);              \
  run<2, F>("mixK", 3, sink, cyc, grid);              \
  run<4, F>("mixK", 3, sink, cyc, grid);              \
  run<6, F>("mixK", 3, sink, cyc, grid);              \
  run<8, F>("mixK", 3, sink, cyc, grid);              \
  run<6, F>("mixK_x2", 4, sink, cyc, grid);           \
  run<4, F>("gapK_both", 5, sink, cyc, grid);
  SWEEP(0) SWEEP(1) SWEEP(2) SWEEP(3) SWEEP(4) SWEEP(5) SWEEP(6) SWEEP(7) SWEEP(8) SWEEP(9)
  return 0;
}
