// pool_eltwise.hip -- the reference's two roadmap ops as HBM-bound element kernels (gfx950):
//   max pooling over NHWC (the pooling stage of conv+relu+pool, README.md:64 and the MKL-DNN
//   pipeline of test/test_conv_relu_pooling.cc:165-226) and eltwise-sum (+ReLU) (README.md:65;
//   the "shortcut sum" post-op of test_conv_relu_pooling.cc:118-124).
// Pooling: dst[n,oy,ox,c] = max over the window positions that lie inside the input; one thread
// per 16 bytes of channels of one output pixel (consecutive threads = consecutive channel groups:
// coalesced reads and writes), scalar path for channel counts that are not a 16-byte multiple.
// Eltwise: dst[i] = relu?(saturate(sum_k src_k[i])), integers summed exactly, f32 left to right.
#include "dfx_device.cuh"

namespace dfx {

template <typename T>
__device__ __forceinline__ T pool_lowest();
template <> __device__ __forceinline__ float pool_lowest<float>() { return -__builtin_inff(); }
template <> __device__ __forceinline__ int pool_lowest<int>() { return (int)0x80000000; }
template <> __device__ __forceinline__ signed char pool_lowest<signed char>() { return (signed char)-128; }
template <> __device__ __forceinline__ unsigned char pool_lowest<unsigned char>() { return 0; }

// f32 maximum with the operand order of vmaxps(acc, x): the second operand wins ties and NaNs
__device__ __forceinline__ float pool_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ int pool_max(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ signed char pool_max(signed char a, signed char b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned char pool_max(unsigned char a, unsigned char b) { return a > b ? a : b; }

template <typename T, int N>  // N elements of T = 16 bytes (vec) or 1 element (scalar)
__device__ __forceinline__ void pool_item(const PoolArgs &a, long long id) {
  const long long px = id / a.groups;
  const int g = (int)(id - px * a.groups);
  const int ox = (int)(px % a.ow);
  const long long t = px / a.ow;
  const int oy = (int)(t % a.oh), n = (int)(t / a.oh);
  T acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = pool_lowest<T>();
  const int y0 = oy * a.sh - a.pad_t, x0 = ox * a.sw - a.pad_l;
  for (int ky = 0; ky < a.kh; ++ky) {
    const int y = y0 + ky;
    if (y < 0 || y >= a.ih) continue;
    for (int kx = 0; kx < a.kw; ++kx) {
      const int x = x0 + kx;
      if (x < 0 || x >= a.iw) continue;
      const T *p = reinterpret_cast<const T *>(a.src) + (((size_t)n * a.ih + y) * a.iw + x) * a.c + (size_t)g * N;
      T v[N];
      if (N * sizeof(T) == 16) *reinterpret_cast<v4i *>(v) = *reinterpret_cast<const v4i *>(p);
      else v[0] = p[0];
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = pool_max(acc[e], v[e]);
    }
  }
  T *q = reinterpret_cast<T *>(a.dst) + (size_t)px * a.c + (size_t)g * N;
  if (N * sizeof(T) == 16) dfx_store16(reinterpret_cast<v4i *>(q), *reinterpret_cast<const v4i *>(acc));
  else q[0] = acc[0];
}

// average pooling: integer types sum exactly (the window is small: int is wide enough for 1-byte types,
// long long for s32), then float(sum) / float(count) -> nearest even -> saturation; f32 sums in window order
template <typename T> struct AvgAcc { typedef int type; };
template <> struct AvgAcc<int> { typedef long long type; };
template <> struct AvgAcc<float> { typedef float type; };
template <typename T>
__device__ __forceinline__ T avg_finish(typename AvgAcc<T>::type sum, int count);
template <> __device__ __forceinline__ float avg_finish<float>(float sum, int count) { return __fdiv_rn(sum, (float)count); }
template <> __device__ __forceinline__ int avg_finish<int>(long long sum, int count) {
  const float q = __builtin_rintf(__fdiv_rn((float)sum, (float)count));
  return q >= 2147483648.0f ? 2147483647 : q <= -2147483648.0f ? (int)0x80000000 : (int)q;
}
template <> __device__ __forceinline__ signed char avg_finish<signed char>(int sum, int count) {
  return (signed char)min(127, max(-128, (int)__builtin_rintf(__fdiv_rn((float)sum, (float)count))));
}
template <> __device__ __forceinline__ unsigned char avg_finish<unsigned char>(int sum, int count) {
  return (unsigned char)min(255, max(0, (int)__builtin_rintf(__fdiv_rn((float)sum, (float)count))));
}

template <typename T, int N>
__device__ __forceinline__ void avgpool_item(const PoolArgs &a, long long id) {
  const long long px = id / a.groups;
  const int g = (int)(id - px * a.groups);
  const int ox = (int)(px % a.ow);
  const long long t = px / a.ow;
  const int oy = (int)(t % a.oh), n = (int)(t / a.oh);
  typename AvgAcc<T>::type acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0;
  const int y0 = oy * a.sh - a.pad_t, x0 = ox * a.sw - a.pad_l;
  int inside = 0;
  for (int ky = 0; ky < a.kh; ++ky) {
    const int y = y0 + ky;
    if (y < 0 || y >= a.ih) continue;
    for (int kx = 0; kx < a.kw; ++kx) {
      const int x = x0 + kx;
      if (x < 0 || x >= a.iw) continue;
      const T *p = reinterpret_cast<const T *>(a.src) + (((size_t)n * a.ih + y) * a.iw + x) * a.c + (size_t)g * N;
      T v[N];
      if (N * sizeof(T) == 16) *reinterpret_cast<v4i *>(v) = *reinterpret_cast<const v4i *>(p);
      else v[0] = p[0];
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = acc[e] + (typename AvgAcc<T>::type)v[e];
      ++inside;
    }
  }
  const int count = a.algo == DFX_POOL_AVG_INCLUDE_PADDING ? a.kh * a.kw : inside;
  T out[N];
#pragma unroll
  for (int e = 0; e < N; ++e) out[e] = avg_finish<T>(acc[e], count);
  T *q = reinterpret_cast<T *>(a.dst) + (size_t)px * a.c + (size_t)g * N;
  if (N * sizeof(T) == 16) dfx_store16(reinterpret_cast<v4i *>(q), *reinterpret_cast<const v4i *>(out));
  else q[0] = out[0];
}

template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(PoolArgs a) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < a.total; id += stride) {
    if (a.algo == DFX_POOL_MAX) {
      if (a.vec) pool_item<T, 16 / (int)sizeof(T)>(a, id);
      else pool_item<T, 1>(a, id);
    } else {
      if (a.vec) avgpool_item<T, 16 / (int)sizeof(T)>(a, id);
      else avgpool_item<T, 1>(a, id);
    }
  }
}

int launch_pool(const PoolArgs &a, hipStream_t s) {
  if (a.total == 0) return 0;
  long long blocks = (a.total + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  switch (a.dt) {
    case DFX_F32: pool_kernel<float><<<(int)blocks, 256, 0, s>>>(a); break;
    case DFX_S32: pool_kernel<int><<<(int)blocks, 256, 0, s>>>(a); break;
    case DFX_S8: pool_kernel<signed char><<<(int)blocks, 256, 0, s>>>(a); break;
    case DFX_U8: pool_kernel<unsigned char><<<(int)blocks, 256, 0, s>>>(a); break;
    default: return -1;
  }
  return 0;
}

// ---- eltwise sum ----
template <typename T> struct EltAcc { typedef int type; };                 // 1-byte types: <= 8 x 255 fits an int
template <> struct EltAcc<int> { typedef long long type; };
template <> struct EltAcc<float> { typedef float type; };

template <typename T>
__device__ __forceinline__ T elt_finish(typename EltAcc<T>::type v, bool relu);
template <> __device__ __forceinline__ float elt_finish<float>(float v, bool relu) { return relu ? relu_x86(v) : v; }
template <> __device__ __forceinline__ int elt_finish<int>(long long v, bool relu) {
  if (relu && v < 0) v = 0;
  return (int)(v < -2147483648LL ? -2147483648LL : (v > 2147483647LL ? 2147483647LL : v));
}
template <> __device__ __forceinline__ signed char elt_finish<signed char>(int v, bool relu) {
  return (signed char)min(127, max(relu ? 0 : -128, v));
}
template <> __device__ __forceinline__ unsigned char elt_finish<unsigned char>(int v, bool) {
  return (unsigned char)min(255, v);  // (sums of u8 are never negative: relu is the identity)
}

template <typename T, int N>  // N elements starting at e0: 16 bytes (N = 16 / sizeof(T)) or one element
__device__ __forceinline__ void eltwise_item(const EltwiseArgs &a, long long e0) {
  typename EltAcc<T>::type acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0;
  for (int k = 0; k < a.n_inputs; ++k) {
    const T *p = reinterpret_cast<const T *>(a.src[k]) + e0;
    T v[N];
    if (N * sizeof(T) == 16) *reinterpret_cast<v4i *>(v) = *reinterpret_cast<const v4i *>(p);
    else v[0] = p[0];
#pragma unroll
    for (int e = 0; e < N; ++e)  // (f32: the first term is taken as it is, then left to right)
      acc[e] = k == 0 ? (typename EltAcc<T>::type)v[e] : acc[e] + (typename EltAcc<T>::type)v[e];
  }
  T out[N];
#pragma unroll
  for (int e = 0; e < N; ++e) out[e] = elt_finish<T>(acc[e], a.relu != 0);
  T *q = reinterpret_cast<T *>(a.dst) + e0;
  if (N * sizeof(T) == 16) dfx_store16(reinterpret_cast<v4i *>(q), *reinterpret_cast<const v4i *>(out));
  else q[0] = out[0];
}

template <typename T>
__global__ __launch_bounds__(256) void eltwise_kernel(EltwiseArgs a) {
  constexpr int N = 16 / (int)sizeof(T);
  const long long nvec = a.elems / N, tail = a.elems - nvec * N, stride = (long long)gridDim.x * blockDim.x;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nvec + tail; id += stride) {
    if (id < nvec) eltwise_item<T, N>(a, id * N);
    else eltwise_item<T, 1>(a, nvec * N + (id - nvec));
  }
}

int launch_eltwise(const EltwiseArgs &a, hipStream_t s) {
  if (a.elems == 0) return 0;
  const long long items = a.elems / (16 / (a.dt == DFX_F32 || a.dt == DFX_S32 ? 4 : 1)) + 16;
  long long blocks = (items + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  switch (a.dt) {
    case DFX_F32: eltwise_kernel<float><<<(int)blocks, 256, 0, s>>>(a); break;
    case DFX_S32: eltwise_kernel<int><<<(int)blocks, 256, 0, s>>>(a); break;
    case DFX_S8: eltwise_kernel<signed char><<<(int)blocks, 256, 0, s>>>(a); break;
    case DFX_U8: eltwise_kernel<unsigned char><<<(int)blocks, 256, 0, s>>>(a); break;
    default: return -1;
  }
  return 0;
}

}  // namespace dfx
