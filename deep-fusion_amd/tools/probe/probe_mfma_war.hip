// probe_mfma_war.hip -- isolated probe of a gfx950 write-after-read hazard on MFMA sources:
// an LDS load that overwrites the SrcA/SrcB registers of a v_mfma issued shortly BEFORE it.
//
// Background: round 1 saw one wrong output element in conv_mfma_fused_kernel<2,2,4,4>
// (gpurun_out/pytest_gpu.log); the disassembly of that build
// (profiles/debug/war_scan.py on gpurun_out/libdfx_hip_prev.so) has 29 pairs
//     v_mfma_i32_32x32x32_i8 D, A, B, C ; ds_read_b128 A|B, ...        (distance 0)
// hipcc's hazard recogniser inserts nothing there.  This probe runs that instruction pair in
// isolation, with a configurable gap between the MFMA and the load, and compares every MFMA
// result with the value the same operands give when nothing is reloaded.
//
//   B<k> / A<k> : the load overwrites SrcB / SrcA of the MFMA issued just before; between them
//                 nothing (k = -1) or "s_nop k" (k + 1 wait states)
//   Bv<k>       : k v_mov_b32 (independent VALU instructions) in the gap instead of s_nop
//   ctl         : same instruction stream, but each load targets a register the preceding MFMA
//                 does NOT read (it was a source of the MFMA before that one)
//   Bm          : one independent MFMA of the same wave between the MFMA and the load
//
// 16 or 4 waves per CU (4 or 1 per SIMD) on every CU; one launch per configuration; registers
// and LDS only, nothing here can fault.  Output: one JSON line per configuration.
//
// build: hipcc -O2 --offload-arch=gfx950 probe_mfma_war.hip -o probe_mfma_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ bool same(v16i a, v16i b) {
  bool s = true;
#pragma unroll
  for (int i = 0; i < 16; ++i) s = s && (a[i] == b[i]);
  return s;
}

struct Setup {
  v4i f0, f1, fo;
  unsigned a0;
};

__device__ __forceinline__ Setup setup(unsigned char *frag, unsigned seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *base = frag + wave * 4096;  // [f][lane][16 B]: f = 0,1 reloaded operand, 2 the other
  unsigned s = seed ^ (blockIdx.x * 977u + threadIdx.x * 131u);
  for (int f = 0; f < 3; ++f) {
    v4i v;
    for (int j = 0; j < 4; ++j) {
      s = s * 1664525u + 1013904223u;
      v[j] = (int)s;
    }
    *reinterpret_cast<v4i *>(base + f * 1024 + lane * 16) = v;
  }
  __syncthreads();
  Setup r;
  r.a0 = (unsigned)(size_t)(base + lane * 16) & 0xffffu;
  r.f0 = *reinterpret_cast<v4i *>(base + lane * 16);
  r.f1 = *reinterpret_cast<v4i *>(base + 1024 + lane * 16);
  r.fo = *reinterpret_cast<v4i *>(base + 2048 + lane * 16);
  return r;
}

// SRCS: operand order of the probed MFMAs; operand numbering inside the asm:
//   %0 d0, %1 d1, %2 x (the reloaded quad), %3..%10 scratch, %11 fo (the other operand), %12 a0
#define PROBE_KERNEL_L(NAME, A_IS_RELOADED, SRCS, GAP, LEAD)                                              \
  __global__ __launch_bounds__(1024) void NAME(int niter, unsigned long long *bad, unsigned seed) { \
    __shared__ __attribute__((aligned(16))) unsigned char frag[16 * 4096];                        \
    const Setup u = setup(frag, seed);                                                            \
    const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                \
    const v16i r0 = A_IS_RELOADED ? __builtin_amdgcn_mfma_i32_32x32x32_i8(u.f0, u.fo, z, 0, 0, 0)  \
                                  : __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f0, z, 0, 0, 0); \
    const v16i r1 = A_IS_RELOADED ? __builtin_amdgcn_mfma_i32_32x32x32_i8(u.f1, u.fo, z, 0, 0, 0)  \
                                  : __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f1, z, 0, 0, 0); \
    unsigned long long nb0 = 0, nb1 = 0, nbx = 0;                                                 \
    for (int it = 0; it < niter; ++it) {                                                          \
      v16i d0, d1;                                                                                \
      v4i x = u.f0;                                                                               \
      int t0, t1, t2, t3, t4, t5, t6, t7;                                                         \
      asm volatile(LEAD                                                                         \
                   "v_mfma_i32_32x32x32_i8 %0, " SRCS ", 0\n\t" GAP                               \
                   "ds_read_b128 %2, %12 offset:1024\n\t"                                         \
                   "s_waitcnt lgkmcnt(0)\n\t"                                                     \
                   "v_mfma_i32_32x32x32_i8 %1, " SRCS ", 0\n\t" GAP                               \
                   "ds_read_b128 %2, %12\n\t"                                                     \
                   "s_waitcnt lgkmcnt(0)\n\t"                                                     \
                   "s_nop 15\n\ts_nop 15\n\t"                                                     \
                   : "=&v"(d0), "=&v"(d1), "+v"(x), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),   \
                     "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)                                   \
                   : "v"(u.fo), "v"(u.a0)                                                         \
                   : "memory");                                                                   \
      if (!same(d0, r0)) ++nb0;                                                                   \
      if (!same(d1, r1)) ++nb1;                                                                   \
      if (!(x[0] == u.f0[0] && x[1] == u.f0[1] && x[2] == u.f0[2] && x[3] == u.f0[3])) ++nbx;    \
    }                                                                                             \
    if (nb0) atomicAdd(bad, nb0);                                                                 \
    if (nb1) atomicAdd(bad + 1, nb1);                                                             \
    if (nbx) atomicAdd(bad + 2, nbx);                                                             \
  }

#define PROBE_KERNEL(NAME, A, SRCS, GAP) PROBE_KERNEL_L(NAME, A, SRCS, GAP, "s_nop 7\n\t")
#define B_SRCS "%11, %2"
#define A_SRCS "%2, %11"
#define VM "v_mov_b32 %3, %12\n\t"
PROBE_KERNEL(k_B_none, false, B_SRCS, "")
PROBE_KERNEL(k_B_nop0, false, B_SRCS, "s_nop 0\n\t")
PROBE_KERNEL(k_B_nop1, false, B_SRCS, "s_nop 1\n\t")
PROBE_KERNEL(k_B_nop3, false, B_SRCS, "s_nop 3\n\t")
PROBE_KERNEL(k_B_nop7, false, B_SRCS, "s_nop 7\n\t")
PROBE_KERNEL(k_B_nop15, false, B_SRCS, "s_nop 15\n\t")
PROBE_KERNEL(k_B_nop31, false, B_SRCS, "s_nop 15\n\ts_nop 15\n\t")
PROBE_KERNEL(k_B_v1, false, B_SRCS, VM)
PROBE_KERNEL(k_B_v2, false, B_SRCS, VM "v_mov_b32 %4, %12\n\t")
PROBE_KERNEL(k_B_v4, false, B_SRCS, VM "v_mov_b32 %4, %12\n\tv_mov_b32 %5, %12\n\tv_mov_b32 %6, %12\n\t")
PROBE_KERNEL(k_B_v8, false, B_SRCS, VM "v_mov_b32 %4, %12\n\tv_mov_b32 %5, %12\n\tv_mov_b32 %6, %12\n\t"
             "v_mov_b32 %7, %12\n\tv_mov_b32 %8, %12\n\tv_mov_b32 %9, %12\n\tv_mov_b32 %10, %12\n\t")
// no wait states between the compiler's own v_mov into the reloaded quad and the first MFMA
// (the compiler cannot see into the asm block, so it inserts nothing): round-2 finding, see DESIGN
PROBE_KERNEL_L(k_B_rawvalu, false, B_SRCS, "", "")
PROBE_KERNEL(k_A_none, true, A_SRCS, "")
PROBE_KERNEL(k_A_nop3, true, A_SRCS, "s_nop 3\n\t")
PROBE_KERNEL(k_A_nop7, true, A_SRCS, "s_nop 7\n\t")

// control: the two loads are swapped between two register quads x, y so that a load never
// targets a source of the MFMA issued just before it, only of the one before that
__global__ __launch_bounds__(1024) void k_ctl(int niter, unsigned long long *bad, unsigned seed) {
  __shared__ __attribute__((aligned(16))) unsigned char frag[16 * 4096];
  const Setup u = setup(frag, seed);
  const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const v16i r0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f0, z, 0, 0, 0);
  const v16i r1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f1, z, 0, 0, 0);
  unsigned long long nb0 = 0, nb1 = 0, nbx = 0;
  for (int it = 0; it < niter; ++it) {
    v16i d0, d1;
    v4i x = u.f0, y = u.f1;
    asm volatile("s_nop 7\n\t"
                 "v_mfma_i32_32x32x32_i8 %0, %4, %2, 0\n\t"   // reads x
                 "ds_read_b128 %3, %5 offset:1024\n\t"        // writes y (not a source of it)
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_i32_32x32x32_i8 %1, %4, %3, 0\n\t"   // reads y
                 "ds_read_b128 %2, %5\n\t"                    // writes x (source of the MFMA before)
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_nop 15\n\ts_nop 15\n\t"
                 : "=&v"(d0), "=&v"(d1), "+v"(x), "+v"(y)
                 : "v"(u.fo), "v"(u.a0)
                 : "memory");
    if (!same(d0, r0)) ++nb0;
    if (!same(d1, r1)) ++nb1;
    if (!(x[0] == u.f0[0] && x[3] == u.f0[3] && y[0] == u.f1[0] && y[3] == u.f1[3])) ++nbx;
  }
  if (nb0) atomicAdd(bad, nb0);
  if (nb1) atomicAdd(bad + 1, nb1);
  if (nbx) atomicAdd(bad + 2, nbx);
}

// one independent MFMA of the same wave between the probed MFMA and the load
__global__ __launch_bounds__(1024) void k_Bm(int niter, unsigned long long *bad, unsigned seed) {
  __shared__ __attribute__((aligned(16))) unsigned char frag[16 * 4096];
  const Setup u = setup(frag, seed);
  const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const v16i r0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f0, z, 0, 0, 0);
  const v16i r1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f1, z, 0, 0, 0);
  unsigned long long nb0 = 0, nb1 = 0, nbx = 0;
  for (int it = 0; it < niter; ++it) {
    v16i d0, d1, d2;
    v4i x = u.f0;
    asm volatile("s_nop 7\n\t"
                 "v_mfma_i32_32x32x32_i8 %0, %4, %2, 0\n\t"
                 "v_mfma_i32_32x32x32_i8 %3, %4, %4, 0\n\t"   // independent, other registers
                 "ds_read_b128 %2, %5 offset:1024\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_mfma_i32_32x32x32_i8 %1, %4, %2, 0\n\t"
                 "v_mfma_i32_32x32x32_i8 %3, %4, %4, 0\n\t"
                 "ds_read_b128 %2, %5\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_nop 15\n\ts_nop 15\n\t"
                 : "=&v"(d0), "=&v"(d1), "+v"(x), "=&v"(d2)
                 : "v"(u.fo), "v"(u.a0)
                 : "memory");
    if (!same(d0, r0)) ++nb0;
    if (!same(d1, r1)) ++nb1;
    if (!(x[0] == u.f0[0] && x[1] == u.f0[1] && x[2] == u.f0[2] && x[3] == u.f0[3])) ++nbx;
  }
  if (nb0) atomicAdd(bad, nb0);
  if (nb1) atomicAdd(bad + 1, nb1);
  if (nbx) atomicAdd(bad + 2, nbx);
}

// VALU write-after-read: a v_mov overwrites one register of SrcA / SrcB of the MFMA issued just
// before it (hipcc emits this shape: the address of the next fragment load is computed into the
// fragment's own first register); the register is restored before the next use
#define VWAR_KERNEL(NAME, SRCS, GAP, RELOADED_IS_A)                                                \
  __global__ __launch_bounds__(1024) void NAME(int niter, unsigned long long *bad, unsigned seed) { \
    __shared__ __attribute__((aligned(16))) unsigned char frag[16 * 4096];                        \
    const Setup u = setup(frag, seed);                                                            \
    const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                \
    const v16i r0 = RELOADED_IS_A ? __builtin_amdgcn_mfma_i32_32x32x32_i8(u.f0, u.fo, z, 0, 0, 0)  \
                                  : __builtin_amdgcn_mfma_i32_32x32x32_i8(u.fo, u.f0, z, 0, 0, 0); \
    unsigned long long nb0 = 0, nb1 = 0;                                                          \
    for (int it = 0; it < niter; ++it) {                                                          \
      v16i d0, d1;                                                                                \
      int t0 = (int)u.a0 * 77 + it;                                                               \
      asm volatile("v_mov_b32 v120, %3\n\tv_mov_b32 v121, %4\n\tv_mov_b32 v122, %5\n\tv_mov_b32 v123, %6\n\t" \
                   "s_nop 7\n\t"                                                                  \
                   "v_mfma_i32_32x32x32_i8 %0, " SRCS ", 0\n\t" GAP                               \
                   "v_mov_b32 v120, %7\n\t"            /* clobbers register 0 of the quad */      \
                   "s_nop 7\n\t"                                                                  \
                   "v_mov_b32 v120, %3\n\t"            /* restore */                              \
                   "s_nop 7\n\t"                                                                  \
                   "v_mfma_i32_32x32x32_i8 %1, " SRCS ", 0\n\t" GAP                               \
                   "v_mov_b32 v120, %7\n\t"                                                       \
                   "s_nop 15\n\ts_nop 15\n\t"                                                    \
                   : "=&v"(d0), "=&v"(d1)                                                         \
                   : "v"(u.fo), "v"(u.f0[0]), "v"(u.f0[1]), "v"(u.f0[2]), "v"(u.f0[3]), "v"(t0)   \
                   : "v120", "v121", "v122", "v123", "memory");                                   \
      if (!same(d0, r0)) ++nb0;                                                                   \
      if (!same(d1, r0)) ++nb1;                                                                   \
    }                                                                                             \
    if (nb0) atomicAdd(bad, nb0);                                                                 \
    if (nb1) atomicAdd(bad + 1, nb1);                                                             \
  }
VWAR_KERNEL(k_vA_none, "v[120:123], %2", "", true)
VWAR_KERNEL(k_vA_nop0, "v[120:123], %2", "s_nop 0\n\t", true)
VWAR_KERNEL(k_vA_nop1, "v[120:123], %2", "s_nop 1\n\t", true)
VWAR_KERNEL(k_vA_nop3, "v[120:123], %2", "s_nop 3\n\t", true)
VWAR_KERNEL(k_vB_none, "%2, v[120:123]", "", false)
VWAR_KERNEL(k_vB_nop0, "%2, v[120:123]", "s_nop 0\n\t", false)
VWAR_KERNEL(k_vB_nop1, "%2, v[120:123]", "s_nop 1\n\t", false)
VWAR_KERNEL(k_vB_nop3, "%2, v[120:123]", "s_nop 3\n\t", false)

typedef void (*kern_t)(int, unsigned long long *, unsigned);
struct Cfg { const char *name; kern_t k; const char *what; };

int main(int argc, char **argv) {
  const int niter = argc > 1 ? atoi(argv[1]) : 100000;
  unsigned long long *bad;
  CK(hipMalloc(&bad, 8 * 4));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int grid = p.multiProcessorCount;
  const Cfg cfgs[] = {
      {"B_none", k_B_none, "load overwrites SrcB, 0 wait states after the MFMA"},
      {"B_nop0", k_B_nop0, "SrcB, s_nop 0 (1 wait state)"},
      {"B_nop1", k_B_nop1, "SrcB, s_nop 1 (2 wait states)"},
      {"B_nop3", k_B_nop3, "SrcB, s_nop 3 (4 wait states)"},
      {"B_nop7", k_B_nop7, "SrcB, s_nop 7 (8 wait states)"},
      {"B_nop15", k_B_nop15, "SrcB, s_nop 15 (16 wait states)"},
      {"B_nop31", k_B_nop31, "SrcB, 2 x s_nop 15 (32 wait states)"},
      {"B_v1", k_B_v1, "SrcB, 1 independent VALU instruction in between"},
      {"B_v2", k_B_v2, "SrcB, 2 VALU"},
      {"B_v4", k_B_v4, "SrcB, 4 VALU"},
      {"B_v8", k_B_v8, "SrcB, 8 VALU"},
      {"B_valu_raw", k_B_rawvalu, "as B_none, but the v_mov that initialises SrcB sits directly in front of the first MFMA"},
      {"A_none", k_A_none, "load overwrites SrcA, 0 wait states"},
      {"A_nop3", k_A_nop3, "SrcA, s_nop 3"},
      {"A_nop7", k_A_nop7, "SrcA, s_nop 7"},
      {"B_mfma", k_Bm, "SrcB, one independent MFMA of the same wave in between"},
      {"valuA_none", k_vA_none, "v_mov overwrites a register of SrcA, 0 wait states after the MFMA"},
      {"valuA_nop0", k_vA_nop0, "VALU write to SrcA, s_nop 0 (1 wait state)"},
      {"valuA_nop1", k_vA_nop1, "VALU write to SrcA, s_nop 1 (2 wait states)"},
      {"valuA_nop3", k_vA_nop3, "VALU write to SrcA, s_nop 3 (4 wait states)"},
      {"valuB_none", k_vB_none, "v_mov overwrites a register of SrcB, 0 wait states after the MFMA"},
      {"valuB_nop0", k_vB_nop0, "VALU write to SrcB, s_nop 0 (1 wait state)"},
      {"valuB_nop1", k_vB_nop1, "VALU write to SrcB, s_nop 1 (2 wait states)"},
      {"valuB_nop3", k_vB_nop3, "VALU write to SrcB, s_nop 3 (4 wait states)"},
      {"ctl", k_ctl, "control: load targets a source of the MFMA BEFORE the preceding one"},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int wps = 4; wps >= 1; wps -= 3)
    for (const Cfg &c : cfgs) {
      CK(hipMemset(bad, 0, 32));
      CK(hipEventRecord(e0));
      c.k<<<grid, 256 * wps>>>(niter, bad, 0x1234567u);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[3];
      CK(hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost));
      printf("{\"probe\": \"mfma_war\", \"cfg\": \"%s\", \"what\": \"%s\", \"waves_per_simd\": %d, \"cus\": %d, "
             "\"mfma_load_pairs\": %.3g, \"wrong_first_mfma\": %llu, \"wrong_second_mfma\": %llu, "
             "\"wrong_reloaded_operand\": %llu, \"ms\": %.1f}\n",
             c.name, c.what, wps, grid, (double)niter * 2 * 4 * wps * grid, h[0], h[1], h[2], ms);
      fflush(stdout);
    }
  return 0;
}
