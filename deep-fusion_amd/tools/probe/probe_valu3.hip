// probe_valu3.hip -- issue cost of the requant epilogue's VALU instructions on gfx950, alone and with
// 2 / 4 waves per SIMD: v_cvt_f32_i32, v_add_f32, v_mul_f32, v_pk_add_f32, v_pk_mul_f32,
// v_cvt_pk_u8_f32, v_add_u32.  16 independent destination registers per instruction kind, unrolled;
// cycles from s_memtime.  build: hipcc -O2 --offload-arch=gfx950 probe_valu3.hip -o probe_valu3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned *out, unsigned long long *cyc, int iters) {
  float f[16];
  v2f p[16];
  int n[16];
  unsigned u[16];
  for (int i = 0; i < 16; ++i) { f[i] = threadIdx.x + i; p[i] = v2f{f[i], f[i] + 1}; n[i] = threadIdx.x * 3 + i; u[i] = i; }
  const float c1 = 1.0001f + threadIdx.x * 1e-9f;
  const v2f c2 = {c1, c1 + 1e-7f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define DO(i)                                                                                         \
    if (KIND == 0) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(n[i]));                    \
    if (KIND == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "v"(c1));                      \
    if (KIND == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "v"(c1));                      \
    if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(c2));                   \
    if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(c2));                   \
    if (KIND == 5) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(f[i]));           \
    if (KIND == 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));         \
    if (KIND == 7) asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(u[i]));                        \
    if (KIND == 8) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(f[i]) : "v"(c1));
    REP16(DO) REP16(DO) REP16(DO) REP16(DO)
#undef DO
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
  for (int i = 0; i < 16; ++i) acc += (unsigned)f[i] + (unsigned)p[i][0] + (unsigned)p[i][1] + n[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned *out; unsigned long long *cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 4096 * 8);
  const int iters = 2000;
  const char *names[] = {"v_cvt_f32_i32", "v_add_f32", "v_mul_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_cvt_pk_u8_f32", "v_add_u32", "v_xor_b32 literal", "v_fma_f32"};
  void (*ks[])(unsigned *, unsigned long long *, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>};
  for (int kind = 0; kind < 9; ++kind)
    for (int wps : {1, 2, 4}) {
      for (int rep = 0; rep < 2; ++rep) { ks[kind]<<<256, 256 * wps>>>(out, cyc, iters); hipDeviceSynchronize(); }
      std::vector<unsigned long long> c(256);
      hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto x : c) avg += x; avg /= 256;
      const double insts = (double)iters * 64;
      printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_wave\": %.2f, \"cycles_per_instr_per_simd\": %.2f}\n",
             names[kind], wps, avg / insts, avg / insts / wps);
    }
  return 0;
}
