// lds_scribble.hip -- debug aid: fills every CU's LDS with a byte pattern so that a
// kernel which reads LDS before writing it shows up (instead of being masked by the
// identical image a previous launch of the same kernel left behind).
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(1024) void k_scribble(unsigned pattern, unsigned *sink) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = pattern + i;
  __syncthreads();
  if (sink && lds[threadIdx.x] == 0x12345678u) sink[0] = 1;
}
extern "C" int dbg_scribble(unsigned pattern, void *stream) {
  hipFuncSetAttribute((const void *)k_scribble, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k_scribble<<<1024, 1024, 160 * 1024, (hipStream_t)stream>>>(pattern, nullptr);
  return (int)hipGetLastError();
}
