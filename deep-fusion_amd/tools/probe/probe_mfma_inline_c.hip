// probe_mfma_inline_c.hip -- does v_mfma_i32_32x32x32_i8 accept a FLOAT inline constant as SrcC and
// treat it as its raw bit pattern?  (hipcc folds a splat of 0x3E22F983 = 1/(2*pi) into the inline
// constant "0.15915494".)  Prints what the hardware returns for (a) that inline constant, (b) the
// same value from 16 VGPRs, (c) inline 1.0, (d) inline integer 7.  The host waits at most 5 s.
// build: hipcc -O2 --offload-arch=gfx950 probe_mfma_inline_c.hip -o probe_mfma_inline_c
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(int *out) {
  v4i a = {0x01010101, 0x01010101, 0x01010101, 0x01010101};  // all ones: every output = K = 32 ... (sum over 32 k of 1*1)
  v4i b = a;
  v16i d;
  if (MODE == 0) asm volatile("s_nop 7\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 0.15915494\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b));
  if (MODE == 1) {
    v16i c;
    for (int i = 0; i < 16; ++i) c[i] = 0x3E22F983 + (int)(threadIdx.x >> 10);  // (not foldable)
    asm volatile("s_nop 7\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %3\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  }
  if (MODE == 2) asm volatile("s_nop 7\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 1.0\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b));
  if (MODE == 3) asm volatile("s_nop 7\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 7\n\ts_nop 15\n\ts_nop 15" : "=&v"(d) : "v"(a), "v"(b));
  for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = d[i];
}

int main() {
  int *out, *h;
  hipHostMalloc((void **)&h, 64 * 16 * 4, hipHostMallocMapped);
  hipHostGetDevicePointer((void **)&out, h, 0);
  const char *names[] = {"inline 0.15915494 (0x3E22F983)", "VGPR splat 0x3E22F983", "inline 1.0 (0x3F800000)", "inline integer 7"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int i = 0; i < 64 * 16; ++i) h[i] = -1;
    if (mode == 0) k<0><<<1, 64>>>(out);
    if (mode == 1) k<1><<<1, 64>>>(out);
    if (mode == 2) k<2><<<1, 64>>>(out);
    if (mode == 3) k<3><<<1, 64>>>(out);
    int waited = 0;
    while (hipStreamQuery(0) == hipErrorNotReady && waited < 50) { usleep(100000); ++waited; }
    const bool done = hipStreamQuery(0) == hipSuccess;
    printf("{\"srcC\": \"%s\", \"kernel_finished\": %s, \"acc[0] lane 0\": \"0x%08x\", \"minus 32 (the dot product: K = 32 of 1 x 1)\": \"0x%08x\"}\n",
           names[mode], done ? "true" : "false", (unsigned)h[0], (unsigned)(h[0] - 32));
    fflush(stdout);
    if (!done) { printf("kernel did not finish: stopping\n"); fflush(stdout); _exit(3); }
  }
  return 0;
}
