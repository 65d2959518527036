// probe_mfma.hip -- calibrates the dense int8 MFMA peak of this GPU ("derive P on
// the GPU box from CU count x clock x MAC/clk with a pure v_mfma_i32_* loop"): every wave issues
// independent v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 chains back to back, 1..4 waves per
// SIMD on every CU; prints achieved TOP/s next to the nominal CUs x 4 SIMDs x 1024 MAC/clk x clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k_mfma(int *out, int iters, int seed) {
  v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed + 4, seed + 5, seed + 6, seed + 7};
  v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
  v4i d0 = {}, d1 = {}, d2 = {}, d3 = {};
  for (int i = 0; i < iters; ++i) {
    if (SHAPE == 32) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
    }
  }
  int s = 0;
  for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
  for (int k = 0; k < 4; ++k) s += d0[k] + d1[k] + d2[k] + d3[k];
  if (s == 0x7fffffff) out[threadIdx.x] = s;  // keep the chains alive
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int cus = p.multiProcessorCount;
  const double clk_ghz = p.clockRate / 1e6;
  const double nominal = cus * 4.0 * 1024 * 2 * clk_ghz / 1e3;  // TOP/s
  printf("%s: %d CUs, max clock %.2f GHz -> nominal dense int8 %.0f TOP/s (1024 MAC/clk/SIMD)\n", p.gcnArchName, cus,
         clk_ghz, nominal);
  int *out;
  (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int shape : {32, 16})
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD
      const int grid = cus * wps;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (shape == 32) k_mfma<32><<<grid, 256>>>(out, iters, rep); else k_mfma<16><<<grid, 256>>>(out, iters, rep);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double macs_per = shape == 32 ? 32.0 * 32 * 32 : 16.0 * 16 * 64;
      const double ops = 2.0 * macs_per * 4 * iters * 4.0 * grid;  // 4 MFMAs/iter, 4 waves/block
      printf("v_mfma_i32_%s_i8, %d wave(s)/SIMD: %8.1f TOP/s  (%.3f ms)\n", shape == 32 ? "32x32x32" : "16x16x64", wps,
             ops / (ms * 1e-3) / 1e12, ms);
    }
  return 0;
}
