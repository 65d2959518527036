// debug_hooks.hip -- test hooks of the C ABI (include/dfx.h, "test hooks").  Not on the hot path.
#include <hip/hip_runtime.h>

#include "../../include/dfx.h"

namespace {
// Fills the whole LDS of every CU with a launch-dependent pattern.  A kernel that reads LDS
// before this launch's own writes have been published then computes from garbage instead of
// from the identical image an earlier launch of the same kernel left behind, which turns a
// read-before-publish race from "rare, first launches only" into a deterministic mismatch.
__global__ __launch_bounds__(1024) void k_scribble_lds(unsigned pattern, unsigned *sink) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = pattern ^ (unsigned)(i * 2654435761u);
  __syncthreads();
  if (sink && lds[threadIdx.x] == 0x12345678u) sink[0] = 1;  // keeps the stores alive
}
}  // namespace

extern "C" int dfx_debug_scribble_lds(unsigned pattern, dfx_stream_t s) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return DFX_ERR_NO_DEVICE;
  if (hipFuncSetAttribute((const void *)k_scribble_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return DFX_ERR_HIP;
  // 4 workgroups per CU's worth of launches: each takes the whole LDS, so every CU runs several
  k_scribble_lds<<<4 * p.multiProcessorCount, 1024, 160 * 1024, (hipStream_t)s>>>(pattern, nullptr);
  return hipGetLastError() == hipSuccess ? DFX_OK : DFX_ERR_HIP;
}
