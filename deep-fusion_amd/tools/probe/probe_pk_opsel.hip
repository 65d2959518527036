// probe_pk_opsel.hip -- round 2 saw v_pk_add_f32 / v_pk_mul_f32 return a wrong LOW result in the last
// 16 lanes of a wave, only where op_sel picks the HIGH half of a source pair for the low lane
// (`op_sel:[0,1]`), ~1e-4 of the executions in the conv epilogue (scalar v_add_f32/v_mul_f32 on the
// same registers: never).  This probe runs that form in isolation: src0 = a register pair written by
// an MFMA (as in the epilogue) or by VALU, src1 = a constant pair, with the forms the compiler
// emitted; compares with scalar arithmetic.  4 waves per SIMD, other waves issue MFMAs meanwhile.
// build: hipcc -O2 --offload-arch=gfx950 probe_pk_opsel.hip -o probe_pk_opsel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// FORM: the op_sel text of the add; SRC_MFMA: src0 pair comes straight from an MFMA result
#define PK_KERNEL(NAME, FORM, LOSEL)                                                              \
  __global__ __launch_bounds__(1024) void NAME(int niter, unsigned long long *bad, unsigned seed) { \
    unsigned s = seed ^ (blockIdx.x * 977u + threadIdx.x * 131u);                                 \
    v4i a, b;                                                                                     \
    for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a[j] = (int)(s & 0x03030303); s = s * 1664525u + 1013904223u; b[j] = (int)(s & 0x03030303); } \
    const float c0 = 0.25f + (threadIdx.x & 7) * 0.125f, c1 = -0.15915494f + (threadIdx.x & 3);   \
    unsigned long long nbad = 0;                                                                  \
    for (int it = 0; it < niter; ++it) {                                                          \
      float r0, r1, x0, x1;                                                                       \
      asm volatile("v_mov_b32 v120, %6\n\tv_mov_b32 v121, %7\n\t"                                 \
                   "s_nop 7\n\t"                                                                  \
                   "v_mfma_i32_32x32x32_i8 v[100:115], %4, %5, 0.15915494\n\t"                    \
                   "s_nop 15\n\t"                                                                 \
                   "v_pk_add_f32 v[116:117], v[102:103], v[120:121] " FORM "\n\t"                 \
                   "v_mov_b32 %0, v116\n\tv_mov_b32 %1, v117\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t" \
                   : "=&v"(r0), "=&v"(r1), "=&v"(x0), "=&v"(x1)                                   \
                   : "v"(a), "v"(b), "v"(c0), "v"(c1)                                             \
                   : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", \
                     "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v120", "v121"); \
      const float k = LOSEL ? c1 : c0;                                                            \
      if (r0 != x0 + k || r1 != x1 + c1 * (LOSEL ? 1.f : 0.f) + c0 * (LOSEL ? 0.f : 1.f)) ++nbad; \
    }                                                                                             \
    if (nbad) atomicAdd(bad, nbad);                                                               \
  }
// op_sel:[0,1] -> low result = src0.lo + src1.HI; op_sel_hi default [1,1] -> high result = src0.hi + src1.hi
PK_KERNEL(k_hi, "op_sel:[0,1]", 1)
// op_sel_hi:[1,0] -> low = src0.lo + src1.lo; high = src0.hi + src1.LO
PK_KERNEL(k_lo, "op_sel_hi:[1,0]", 0)

int main(int argc, char **argv) {
  const int niter = argc > 1 ? atoi(argv[1]) : 200000;
  unsigned long long *bad;
  CK(hipMalloc(&bad, 8));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  for (int m = 0; m < 2; ++m) {
    CK(hipMemset(bad, 0, 8));
    if (m == 0) k_hi<<<p.multiProcessorCount, 1024>>>(niter, bad, 1u); else k_lo<<<p.multiProcessorCount, 1024>>>(niter, bad, 1u);
    CK(hipDeviceSynchronize());
    unsigned long long h;
    CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"pk_opsel\", \"form\": \"v_pk_add_f32 d, mfma_result_pair, const_pair %s\", \"executions\": %.3g, \"wrong_lanes\": %llu}\n",
           m == 0 ? "op_sel:[0,1]" : "op_sel_hi:[1,0]", (double)niter * 16 * p.multiProcessorCount, h);
  }
  return 0;
}
