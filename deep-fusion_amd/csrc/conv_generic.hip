// conv_generic.hip -- shape-general conv / fused conv kernel (gfx950).
//
// Covers every shape the reference's init_conf accepts in principle
// (/root/reference/src/jit_conv_kernel.cc:586-592: ic, oc multiples of 16; any
// kernel size, stride, top/left padding; fused or unfused; the four dst dtypes)
// directly on the reference's own tensor formats: NHWC u8 activations and
// OIhw4i16o4i s8 weights (layout restated from jit_conv_kernel.cc:333-338), so
// no repack is needed.  It is the correctness backstop behind the MFMA variant
// (conv_mfma.cuh), not the roofline kernel.
//
// Work split: one workgroup = TP consecutive output pixels of the flattened
// (n, oh, ow) index.  Phase 1: threads own (pixel, oc) pairs, accumulate the
// exact s32 conv0 result and requantise; fused ops park the u8 intermediate in
// LDS (it never reaches HBM).  Phase 2: threads own (pixel, oc1x1) pairs.
// Consecutive threads own consecutive channels, so weight loads are 64-B
// contiguous per 16 lanes and dst stores are fully coalesced.
#include "dfx_device.cuh"

namespace dfx {

constexpr int GEN_TP = 32;       // pixels per workgroup
constexpr int GEN_THREADS = 256;

template <typename T>
__device__ __forceinline__ void store_typed(T *dst, size_t idx, float f, int rm);
template <>
__device__ __forceinline__ void store_typed<float>(float *dst, size_t idx, float f, int) {
  dst[idx] = f;
}
template <>
__device__ __forceinline__ void store_typed<int>(int *dst, size_t idx, float f, int rm) {
  dst[idx] = cvt_x86_rt(f, rm);
}
template <>
__device__ __forceinline__ void store_typed<int8_t>(int8_t *dst, size_t idx, float f, int rm) {
  dst[idx] = (int8_t)sat_s8(cvt_x86_rt(f, rm));
}
template <>
__device__ __forceinline__ void store_typed<uint8_t>(uint8_t *dst, size_t idx, float f, int rm) {
  dst[idx] = (uint8_t)sat_u8_bits(cvt_x86_rt(f, rm));
}

template <typename DST>
__global__ __launch_bounds__(GEN_THREADS) void conv_generic_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint8_t *mid = smem;  // [GEN_TP][oc] u8 intermediate (fused only)

  const long long total_px = (long long)a.bs * a.oh * a.ow;
  const long long px0 = (long long)blockIdx.x * GEN_TP;
  const int npx = (int)min((long long)GEN_TP, total_px - px0);
  const int nb_ic = a.ic / 16;
  const bool fused = a.oc1 > 0;

  const float *bias0 = a.consts + a.oc;
  const float *scale0 = a.consts + 2 * a.oc;
  const float *bias1 = a.consts + 3 * a.oc + a.oc1;
  const float *scale1 = a.consts + 3 * a.oc + 2 * a.oc1;
  DST *dst = reinterpret_cast<DST *>(a.dst);

  // ---- phase 1: conv0 ----
  for (int idx = threadIdx.x; idx < npx * a.oc; idx += GEN_THREADS) {
    const int p = idx / a.oc, o = idx - p * a.oc;
    const long long g = px0 + p;
    const int n = (int)(g / ((long long)a.oh * a.ow));
    const int rem = (int)(g - (long long)n * a.oh * a.ow);
    const int oy = rem / a.ow, ox = rem - oy * a.ow;
    const uint8_t *src_n = a.src + (size_t)n * a.ih * a.iw * a.ic;
    const int ocb = o >> 4, ol = o & 15;
    int acc = 0;
    for (int kh = 0; kh < a.kh; ++kh) {
      const int iy = oy * a.sh - a.pt + kh;
      if (iy < 0 || iy >= a.ih) continue;  // zero padding = tap skipped (op_conv.cc:218-220)
      for (int kw = 0; kw < a.kw; ++kw) {
        const int ix = ox * a.sw - a.pl + kw;
        if (ix < 0 || ix >= a.iw) continue;  // jit_conv_kernel.h:120-127
        const uint32_t *px =
            reinterpret_cast<const uint32_t *>(src_n + ((size_t)iy * a.iw + ix) * a.ic);
        for (int icb = 0; icb < nb_ic; ++icb) {
          const int *w = reinterpret_cast<const int *>(
              a.wei + (((size_t)ocb * nb_ic + icb) * a.kh * a.kw + (size_t)kh * a.kw + kw) * 256);
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) acc += dot4_u8s8(px[icb * 4 + i4], w[i4 * 16 + ol]);
        }
      }
    }
    if (fused) {
      const float f = requant(acc, bias0[o], scale0[o], true);
      mid[p * a.oc + o] = (uint8_t)sat_u8_bits(cvt_x86_rt(f, a.rm0));
    } else {
      const bool relu = a.relu0 || a.dst_dt == DFX_U8;
      const float f = requant(acc, bias0[o], scale0[o], relu);
      store_typed<DST>(dst, (size_t)g * a.oc + o, f, a.rm0);
    }
  }
  if (!fused) return;
  __syncthreads();

  // ---- phase 2: 1x1 conv on the LDS-resident intermediate ----
  const int nb_oc = a.oc / 16;
  const bool relu1 = a.relu1 || a.dst_dt == DFX_U8;
  for (int idx = threadIdx.x; idx < npx * a.oc1; idx += GEN_THREADS) {
    const int p = idx / a.oc1, o1 = idx - p * a.oc1;
    const uint32_t *m = reinterpret_cast<const uint32_t *>(mid + p * a.oc);
    const int ob = o1 >> 4, ol = o1 & 15;
    int acc = 0;
    for (int k = 0; k < nb_oc; ++k) {
      const int *w = reinterpret_cast<const int *>(a.wei1 + ((size_t)ob * nb_oc + k) * 256);
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) acc += dot4_u8s8(m[k * 4 + i4], w[i4 * 16 + ol]);
    }
    const float f = requant(acc, bias1[o1], scale1[o1], relu1);
    store_typed<DST>(dst, (size_t)(px0 + p) * a.oc1 + o1, f, a.rm1);
  }
}

int launch_conv_generic(const ConvArgs &a, hipStream_t s, int *grid_out, int *lds_out) {
  const long long total_px = (long long)a.bs * a.oh * a.ow;
  const int grid = (int)((total_px + GEN_TP - 1) / GEN_TP);
  const int lds = a.oc1 > 0 ? GEN_TP * a.oc : 0;
  if (grid_out) *grid_out = grid;
  if (lds_out) *lds_out = lds;
  if (!a.src) return 0;  // query only
  switch (a.dst_dt) {
    case DFX_F32: conv_generic_kernel<float><<<grid, GEN_THREADS, lds, s>>>(a); break;
    case DFX_S32: conv_generic_kernel<int><<<grid, GEN_THREADS, lds, s>>>(a); break;
    case DFX_S8: conv_generic_kernel<int8_t><<<grid, GEN_THREADS, lds, s>>>(a); break;
    case DFX_U8: conv_generic_kernel<uint8_t><<<grid, GEN_THREADS, lds, s>>>(a); break;
    default: return -1;
  }
  return 0;
}

}  // namespace dfx
