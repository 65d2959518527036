// Probe: HBM write rate of the s32/f32 epilogue's store patterns on gfx950 (test infrastructure,
// not shipped).  One persistent 1024-thread workgroup per CU; every wave writes 32-pixel x 256-channel
// x 4-byte tiles (32 KB, pixel stride 1 KB) the way one epilogue variant would:
//   A  MFMA(weights, pixels) layout: per instruction each pixel gets 2 lanes x 16 B = 32 B contiguous
//   B  MFMA(pixels, weights) layout: dword stores, 32 lanes = one full 128-byte line of a pixel
//   C  LDS-transposed: dwordx4 stores, 8 lanes = one full 128-byte line
//   D  linear 32 KB (memset-like upper bound)
// Build: hipcc -O2 --offload-arch=gfx950 probe_store_pattern.hip -o probe_store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(1024) void store_kernel(char *dst, int ntiles, int waves_used) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= waves_used) return;
  const int n = lane & 31, h = lane >> 5;
  const int stride = gridDim.x * waves_used;
  for (int t = blockIdx.x * waves_used + wave; t < ntiles; t += stride) {
    char *tile = dst + (size_t)t * 32768;
    v4i v = {t, lane, t ^ lane, 7};
    if (PAT == 0) {
#pragma unroll
      for (int cg = 0; cg < 8; ++cg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v4i *p = (v4i *)(tile + n * 1024 + cg * 128 + j * 32 + h * 16);
          if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    } else if (PAT == 1) {
#pragma unroll
      for (int cg = 0; cg < 8; ++cg)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int px = 8 * (i / 4) + 4 * h + (i % 4);
          int *p = (int *)(tile + px * 1024 + cg * 128 + n * 4);
          if (NT) __builtin_nontemporal_store(v.x + i, p); else *p = v.x + i;
        }
    } else if (PAT == 2) {
#pragma unroll
      for (int cg = 0; cg < 8; ++cg)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v4i *p = (v4i *)(tile + (j * 8 + (lane >> 3)) * 1024 + cg * 128 + (lane & 7) * 16);
          if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        v4i *p = (v4i *)(tile + k * 1024 + lane * 16);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
      }
    }
  }
}

template <int PAT, bool NT>
static void run(const char *name, char *buf, int ntiles, int waves) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) store_kernel<PAT, NT><<<256, 1024>>>(buf, ntiles, waves);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) store_kernel<PAT, NT><<<256, 1024>>>(buf, ntiles, waves);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, gbs = (double)ntiles * 32768 / (us * 1e-6) / 1e9;
  printf("{\"pattern\": \"%s\", \"nt\": %d, \"waves\": %d, \"us\": %.2f, \"GBps\": %.0f}\n", name, NT ? 1 : 0, waves, us, gbs);
}

int main() {
  const int ntiles = 12544;  // 128 x 56 x 56 pixels / 32
  char *buf; hipMalloc(&buf, (size_t)ntiles * 32768);
  for (int waves : {14, 16}) {
    run<0, false>("A_pix32B_x4", buf, ntiles, waves);  run<0, true>("A_pix32B_x4", buf, ntiles, waves);
    run<1, false>("B_line128_dword", buf, ntiles, waves); run<1, true>("B_line128_dword", buf, ntiles, waves);
    run<2, false>("C_line128_x4", buf, ntiles, waves); run<2, true>("C_line128_x4", buf, ntiles, waves);
    run<3, false>("D_linear", buf, ntiles, waves); run<3, true>("D_linear", buf, ntiles, waves);
  }
  hipFree(buf);
  return 0;
}
