// conv_pw.hip -- instantiations of the pointwise (1x1) unfused conv kernel (conv_pw.cuh) and its launcher.
#include "conv_pw.cuh"

namespace dfx {

// mode 0: launch; mode 1: raise the dynamic-LDS limit; mode 2: resident workgroups per CU
template <int OCB, int DST>
static int pw_one(const ConvArgs &a, const PwGeom &g, int grid, int lds, hipStream_t s, int mode) {
  auto k = conv_pw_kernel<OCB, DST>;
  if (mode == 1)
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (mode == 2) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, PW_THREADS, lds) != hipSuccess) return -1;
    return n;
  }
  k<<<grid, PW_THREADS, lds, s>>>(a, g);
  return 0;
}

template <int DST>
static int pw_dst(const ConvArgs &a, const PwGeom &g, int grid, int lds, hipStream_t s, int mode) {
  switch (g.ocb) {
    case 2: return pw_one<2, DST>(a, g, grid, lds, s, mode);
    case 4: return pw_one<4, DST>(a, g, grid, lds, s, mode);
    case 8: return pw_one<8, DST>(a, g, grid, lds, s, mode);
  }
  return -1;
}

int launch_conv_pw(const ConvArgs &a, const PwGeom &g, int dst_dt, int grid, int lds, hipStream_t s, int mode) {
  switch (dst_dt) {
    case DFX_F32: return pw_dst<DFX_F32>(a, g, grid, lds, s, mode);
    case DFX_S32: return pw_dst<DFX_S32>(a, g, grid, lds, s, mode);
    case DFX_S8: return pw_dst<DFX_S8>(a, g, grid, lds, s, mode);
    case DFX_U8: return pw_dst<DFX_U8>(a, g, grid, lds, s, mode);
  }
  return -1;
}

}  // namespace dfx
