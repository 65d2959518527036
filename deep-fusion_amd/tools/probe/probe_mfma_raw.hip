// probe_mfma_raw.hip -- how many wait states does gfx950 need between v_mfma_i32_32x32x32_i8 and a
// VALU instruction that READS its result?  hipcc pads this read-after-write to 12 wait states and the
// hardware does not interlock it.  Round 2 saw rare wrong accumulator values in one 16-lane quarter
// of one register once v_pk_add_f32 read MFMA results directly, so the probe measures the padding
// each consumer needs: v_mov_b32, v_add_f32, v_pk_add_f32 (64-bit operand), reading the FIRST or the
// LAST register (pair) of the 16-register result, K = 0..16 extra wait states, 4 waves per SIMD on
// every CU with all of them issuing MFMAs.  Counts wrong reads.  Registers only: cannot fault.
// build: hipcc -O2 --offload-arch=gfx950 probe_mfma_raw.hip -o probe_mfma_raw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// CONS: 0 v_mov_b32 of register IDX, 1 v_add_f32 (+0.0 in bits: use v_add_u32 with 0 to keep bits) , 2 v_pk_add_f32 of pair IDX..IDX+1
#define RAW_KERNEL(NAME, NOPS, CONSUME)                                                           \
  __global__ __launch_bounds__(1024) void NAME(int niter, unsigned long long *bad, unsigned seed) { \
    unsigned s = seed ^ (blockIdx.x * 977u + threadIdx.x * 131u);                                 \
    v4i a, b;                                                                                     \
    for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a[j] = (int)s; s = s * 1664525u + 1013904223u; b[j] = (int)s; } \
    const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                \
    const v16i ref = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);                     \
    unsigned long long nbad = 0;                                                                  \
    for (int it = 0; it < niter; ++it) {                                                          \
      v16i d = {7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7};                                  \
      v2f lo, hi;                                                                                 \
      asm volatile("s_nop 7\n\t"                                                                  \
                   "v_mfma_i32_32x32x32_i8 %0, %3, %4, 0\n\t" NOPS CONSUME                        \
                   "s_nop 15\n\ts_nop 15\n\t"                                                     \
                   : "+v"(d), "=&v"(lo), "=&v"(hi)                                                \
                   : "v"(a), "v"(b)                                                               \
                   : "memory");                                                                   \
      if (__float_as_int(lo[0]) != ref[0] || __float_as_int(lo[1]) != ref[1]) ++nbad;            \
      if (__float_as_int(hi[0]) != ref[14] || __float_as_int(hi[1]) != ref[15]) ++nbad;          \
    }                                                                                             \
    if (nbad) atomicAdd(bad, nbad);                                                               \
  }
// consumers: lo <- d[0:1], hi <- d[14:15].  %0 is a 16-register tuple: the assembler has no sub-register
// syntax for operands, so the tuple is pinned to v[64:79] through fixed copies below.
#undef RAW_KERNEL
#define RAW_KERNEL(NAME, NOPS, CONSUME)                                                           \
  __global__ __launch_bounds__(1024) void NAME(int niter, unsigned long long *bad, unsigned seed) { \
    unsigned s = seed ^ (blockIdx.x * 977u + threadIdx.x * 131u);                                 \
    v4i a, b;                                                                                     \
    for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a[j] = (int)s; s = s * 1664525u + 1013904223u; b[j] = (int)s; } \
    const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                \
    const v16i ref = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);                     \
    unsigned long long nbad = 0;                                                                  \
    for (int it = 0; it < niter; ++it) {                                                          \
      int l0, l1, h0, h1;                                                                         \
      asm volatile("v_mov_b32 v100, 7\n\tv_mov_b32 v101, 7\n\tv_mov_b32 v114, 7\n\tv_mov_b32 v115, 7\n\t" \
                   "s_nop 7\n\t"                                                                  \
                   "v_mfma_i32_32x32x32_i8 v[100:115], %4, %5, 0\n\t" NOPS CONSUME                \
                   "s_nop 15\n\ts_nop 15\n\t"                                                     \
                   "v_mov_b32 %0, v116\n\tv_mov_b32 %1, v117\n\tv_mov_b32 %2, v118\n\tv_mov_b32 %3, v119\n\t" \
                   : "=&v"(l0), "=&v"(l1), "=&v"(h0), "=&v"(h1)                                   \
                   : "v"(a), "v"(b)                                                               \
                   : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", \
                     "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"); \
      if (l0 != ref[0] || l1 != ref[1]) ++nbad;                                                   \
      if (h0 != ref[14] || h1 != ref[15]) ++nbad;                                                 \
    }                                                                                             \
    if (nbad) atomicAdd(bad, nbad);                                                               \
  }
#define C_MOV "v_mov_b32 v116, v100\n\tv_mov_b32 v117, v101\n\tv_mov_b32 v118, v114\n\tv_mov_b32 v119, v115\n\t"
#define C_PK  "v_pk_add_f32 v[116:117], v[100:101], 0 op_sel_hi:[1,0]\n\tv_pk_add_f32 v[118:119], v[114:115], 0 op_sel_hi:[1,0]\n\t"
#define C_PKH "v_pk_add_f32 v[118:119], v[114:115], 0 op_sel_hi:[1,0]\n\tv_pk_add_f32 v[116:117], v[100:101], 0 op_sel_hi:[1,0]\n\t"
#define N(k) "s_nop " #k "\n\t"
RAW_KERNEL(k_mov_0, "", C_MOV)
RAW_KERNEL(k_mov_4, N(3), C_MOV)
RAW_KERNEL(k_mov_8, N(7), C_MOV)
RAW_KERNEL(k_mov_10, N(9), C_MOV)
RAW_KERNEL(k_mov_11, N(10), C_MOV)
RAW_KERNEL(k_mov_12, N(11), C_MOV)
RAW_KERNEL(k_mov_13, N(12), C_MOV)
RAW_KERNEL(k_mov_14, N(13), C_MOV)
RAW_KERNEL(k_pk_8, N(7), C_PK)
RAW_KERNEL(k_pk_10, N(9), C_PK)
RAW_KERNEL(k_pk_11, N(10), C_PK)
RAW_KERNEL(k_pk_12, N(11), C_PK)
RAW_KERNEL(k_pk_13, N(12), C_PK)
RAW_KERNEL(k_pk_14, N(13), C_PK)
RAW_KERNEL(k_pk_16, N(15), C_PK)
RAW_KERNEL(k_pkh_11, N(10), C_PKH)
RAW_KERNEL(k_pkh_12, N(11), C_PKH)
RAW_KERNEL(k_pkh_13, N(12), C_PKH)
RAW_KERNEL(k_pkh_14, N(13), C_PKH)

typedef void (*kern_t)(int, unsigned long long *, unsigned);
struct Cfg { const char *name; kern_t k; int states; };
int main(int argc, char **argv) {
  const int niter = argc > 1 ? atoi(argv[1]) : 100000;
  unsigned long long *bad;
  CK(hipMalloc(&bad, 8));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int grid = p.multiProcessorCount;
  const Cfg cfgs[] = {{"v_mov_b32", k_mov_0, 0}, {"v_mov_b32", k_mov_4, 4}, {"v_mov_b32", k_mov_8, 8}, {"v_mov_b32", k_mov_10, 10},
                      {"v_mov_b32", k_mov_11, 11}, {"v_mov_b32", k_mov_12, 12}, {"v_mov_b32", k_mov_13, 13}, {"v_mov_b32", k_mov_14, 14},
                      {"v_pk_add_f32 (first pair read first)", k_pk_8, 8}, {"v_pk_add_f32 (first pair read first)", k_pk_10, 10},
                      {"v_pk_add_f32 (first pair read first)", k_pk_11, 11}, {"v_pk_add_f32 (first pair read first)", k_pk_12, 12},
                      {"v_pk_add_f32 (first pair read first)", k_pk_13, 13}, {"v_pk_add_f32 (first pair read first)", k_pk_14, 14},
                      {"v_pk_add_f32 (first pair read first)", k_pk_16, 16},
                      {"v_pk_add_f32 (last pair read first)", k_pkh_11, 11}, {"v_pk_add_f32 (last pair read first)", k_pkh_12, 12},
                      {"v_pk_add_f32 (last pair read first)", k_pkh_13, 13}, {"v_pk_add_f32 (last pair read first)", k_pkh_14, 14}};
  for (int wps = 4; wps >= 1; wps -= 3)
    for (const Cfg &c : cfgs) {
      CK(hipMemset(bad, 0, 8));
      c.k<<<grid, 256 * wps>>>(niter, bad, 0x1234567u);
      CK(hipDeviceSynchronize());
      unsigned long long h;
      CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
      printf("{\"probe\": \"mfma_raw\", \"consumer\": \"%s\", \"wait_states_after_mfma\": %d, \"waves_per_simd\": %d, "
             "\"mfma_reads\": %.3g, \"wrong_reads\": %llu}\n", c.name, c.states, wps, (double)niter * 2 * 4 * wps * grid, h);
      fflush(stdout);
    }
  return 0;
}
