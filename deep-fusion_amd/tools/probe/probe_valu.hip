// probe_valu.hip -- hardware probes used while designing the requant epilogue:
//  1. rounding/saturation behaviour of v_cvt_pk_u8_f32 (is it usable for the
//     RNE + unsigned-saturate + pack step?)
//  2. VALU issue rate per SIMD at 1, 2 and 4 waves per SIMD (cycles per wave-instruction)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_cvt(const float *in, unsigned *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rate(float *out, long long *cyc, int iters) {
  float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3, g = a + 4, h = a + 5;
  __syncthreads();
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a = __fmul_rn(__fadd_rn(a, c), b); d = __fmul_rn(__fadd_rn(d, c), b);
      e = __fmul_rn(__fadd_rn(e, c), b); f = __fmul_rn(__fadd_rn(f, c), b);
      g = __fmul_rn(__fadd_rn(g, c), b); h = __fmul_rn(__fadd_rn(h, c), b);
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  {
    std::vector<float> v = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 255.4f, 255.5f, 256.f, 1e9f,
                            -0.4f, -0.5f, -0.6f, -3.f, NAN, INFINITY, -INFINITY, 100.49f, 100.5f, 101.5f, 0.49999997f};
    float *din; unsigned *dout; int n = (int)v.size();
    hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(din, v.data(), n * 4, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(din, dout, n);
    std::vector<unsigned> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) = %u   rne+sat = %d\n", v[i], o[i],
                                       std::isnan(v[i]) ? -1 : (int)fmin(255.0, fmax(0.0, nearbyint((double)v[i]))));
  }
  float *out; long long *cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 4096 * 8);
  const int iters = 2000;
  auto run = [&](auto kern, int waves, const char *name) {
    kern<<<256, waves * 64>>>(out, cyc, iters);   // one block per CU
    hipDeviceSynchronize();
    kern<<<256, waves * 64>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> c(256);
    hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : c) avg += x; avg /= 256;
    double insts = (double)iters * 8 * 12;  // VALU per wave
    printf("%s: %d waves/CU (%g per SIMD): %.2f clock64 ticks per wave-instruction per wave, %.2f per SIMD-instruction\n",
           name, waves, waves / 4.0, avg / insts, avg / (insts * waves / 4.0));
  };
  run(k_rate<4>, 4, "rate");
  run(k_rate<8>, 8, "rate");
  run(k_rate<16>, 16, "rate");
  return 0;
}
