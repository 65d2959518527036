// concat.hip -- channel concat of NHWC tensors with optional ReLU (gfx950).
//
// Replaces op_concat<T>::infer + jit_concat_kernel
// (/root/reference/src/op_concat.cc:22-72, src/jit_concat_kernel.cc:30-128):
//   dst[p, off_i + c] = relu?(src_i[p, c])      p = flattened (n, h, w)
// The reference's block rule (every C_i divisible by 16 for 1-byte types, by 4
// for 4-byte types, jit_concat_kernel.cc:155-196) makes every channel span a
// whole number of 16-byte chunks, 16-byte aligned in both source and
// destination, so the kernel moves one 16-byte chunk per lane: consecutive
// lanes write consecutive chunks of dst (fully coalesced, HBM-bound).
// ReLU is the true per-element max(0, x) in the element's own type -- the
// semantics test/test_concat.cc:31-87 pins -- not the reference's vpmaxsw /
// signed-vpmaxsb defects.  f32 follows vmaxps(zero, x).
#include "dfx_device.cuh"

namespace dfx {

__device__ __forceinline__ int relu_dword(int v, int dt) {
  if (dt == DFX_F32) return __float_as_int(relu_x86(__int_as_float(v)));
  if (dt == DFX_S32) return max(v, 0);
  if (dt == DFX_S8) {
    const unsigned neg = ((unsigned)v & 0x80808080u) >> 7;  // 1 in each negative byte
    return (int)((unsigned)v & ~(neg * 0xffu));
  }
  return v;  // u8
}

__global__ __launch_bounds__(256) void concat_kernel(ConcatArgs a) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < a.total_chunks;
       id += stride) {
    const long long p = id / a.chunks_per_px;
    const int k = (int)(id - p * a.chunks_per_px);
    int i = 0, begin = 0;
    while (k >= a.chunk_end[i]) begin = a.chunk_end[i++];
    const int ci = a.chunk_end[i] - begin;  // chunks per pixel of input i
    v4i v = *reinterpret_cast<const v4i *>(a.src[i] + ((size_t)p * ci + (k - begin)) * 16);
    if (a.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = relu_dword(v[e], a.dt);
    }
    dfx_store16(reinterpret_cast<v4i *>(a.dst + (size_t)id * 16), v);
  }
}

int launch_concat(const ConcatArgs &a, hipStream_t s) {
  if (a.total_chunks == 0) return 0;
  long long blocks = (a.total_chunks + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  concat_kernel<<<(int)blocks, 256, 0, s>>>(a);
  return 0;
}

}  // namespace dfx
