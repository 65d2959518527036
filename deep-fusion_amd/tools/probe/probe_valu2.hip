// probe_valu2.hip -- VALU issue throughput per SIMD on gfx950 with 1..8 waves per SIMD,
// integer ops (not packable), measured with s_memtime; also wall time via events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k_rate(unsigned *out, unsigned long long *cyc, int iters) {
  unsigned a = threadIdx.x, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3, e = a ^ 0x55, f = a + 77, g = a * 11, h = a | 3;
  float fa = a, fb = b, fc = c, fd = d;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {  // 8 independent v_add/v_xor chains (int)
        a = (a ^ 0x9e3779b9u) + i; b = (b ^ 0x7f4a7c15u) + i; c = (c ^ 0x85ebca6bu) + i; d = (d ^ 0xc2b2ae35u) + i;
        e = (e ^ 0x27d4eb2fu) + i; f = (f ^ 0x165667b1u) + i; g = (g ^ 0xd3a2646cu) + i; h = (h ^ 0xfd7046c5u) + i;
      } else {          // f32 cvt/add/mul mix like the requant epilogue
        fa = __fmul_rn(__fadd_rn(__int2float_rn((int)a), fa), 1.0001f); a += 3;
        fb = __fmul_rn(__fadd_rn(__int2float_rn((int)b), fb), 1.0001f); b += 5;
        fc = __fmul_rn(__fadd_rn(__int2float_rn((int)c), fc), 1.0001f); c += 7;
        fd = __fmul_rn(__fadd_rn(__int2float_rn((int)d), fd), 1.0001f); d += 9;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (unsigned)(fa + fb + fc + fd);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned *out; unsigned long long *cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int mode = 0; mode < 2; ++mode)
    for (int waves : {4, 8, 16, 32}) {
      auto launch = [&]() {
        if (mode == 0) k_rate<0><<<256, waves * 64 > 1024 ? 1024 : waves * 64>>>(out, cyc, iters);
        else k_rate<1><<<256, waves * 64 > 1024 ? 1024 : waves * 64>>>(out, cyc, iters);
      };
      int blocks_per_cu = waves > 16 ? 2 : 1;
      auto launch2 = [&]() {
        int thr = waves > 16 ? 1024 : waves * 64;
        if (mode == 0) k_rate<0><<<256 * blocks_per_cu, thr>>>(out, cyc, iters);
        else k_rate<1><<<256 * blocks_per_cu, thr>>>(out, cyc, iters);
      };
      launch2(); hipDeviceSynchronize();
      hipEventRecord(e0); launch2(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> c(256);
      hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto x : c) avg += x; avg /= 256;
      // instruction counts per wave per iteration (from the source): mode0: 8*16=128 VALU; mode1: 8*4*4=128 VALU
      double insts = (double)iters * 128;
      printf("mode %d  %2d waves/CU (%4.1f/SIMD): s_memtime ticks per wave-instr %.2f  -> per SIMD-instr %.2f ; wall %.1f us -> %.2f ns per SIMD-instr\n",
             mode, waves, waves / 4.0, avg / insts, avg / insts / (waves / 4.0), ms * 1e3, ms * 1e6 / (insts * waves / 4.0));
    }
  return 0;
}
