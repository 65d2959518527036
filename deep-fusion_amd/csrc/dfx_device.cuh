// dfx_device.cuh -- device-side helpers shared by the conv kernels (gfx950 only).
//
// The requantisation chain reproduces, op for op, what the reference JIT emits
// (/root/reference/src/jit_conv_kernel.cc:256-277 stage 0, :88-130 stage 1):
//   vcvtdq2ps -> vaddps(bias) -> vmulps(scale) -> vmaxps(zero, .) ->
//   vcvtps2dq{rn|rd}-sae -> vpmovusdb | vpmovsdb | raw store
// Each f32 step is a separately rounded IEEE operation (never an FMA), and the
// x86 corner cases are kept: vmaxps returns its second source unless zero > v
// (so -0.0 and NaN pass), vcvtps2dq yields 0x80000000 for NaN / out-of-range.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dfx.h"

namespace dfx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- 16-byte global stores.
// gfx950 (ROCm 7.2): a VALU instruction that overwrites a data register of a global_store_dwordx4
// ONE wait state after the store -- the padding hipcc uses -- can still change what is stored when
// the memory pipeline is backed up (tools/probe/probe_store_war.hip: ~0.5 % of store-bound stores
// wrong at 1 wait state, none at 2; narrower stores and LDS stores are interlocked).  Every 16-byte
// global store of this library goes through dfx_store16*: the asm statement after the store keeps
// the data registers live (an input operand) and is itself 2 wait states, so no instruction can
// overwrite them any earlier.  tests/test_isa_hazards_cpu.py checks the built library for it. ----
template <class V>
__device__ __forceinline__ void dfx_store16(V *p, V v) {
  static_assert(sizeof(V) == 16, "16-byte vector types only");
  *p = v;
  asm volatile("s_nop 1" ::"v"(v) : "memory");
}
template <class V>
__device__ __forceinline__ void dfx_store16_nt(V *p, V v) {  // non-temporal: written once, never re-read here
  static_assert(sizeof(V) == 16, "16-byte vector types only");
  __builtin_nontemporal_store(v, p);
  asm volatile("s_nop 1" ::"v"(v) : "memory");
}

// vmaxps(dst, zero, v)
__device__ __forceinline__ float relu_x86(float v) { return (0.0f > v) ? 0.0f : v; }

// vcvtps2dq with static rounding rn-sae (RM == 0) or rd-sae (RM == 1)
template <int RM>
__device__ __forceinline__ int cvt_x86(float f) {
  float r = RM ? __builtin_floorf(f) : __builtin_rintf(f);  // v_floor_f32 / v_rndne_f32
  int i = (int)r;                                           // v_cvt_i32_f32 (exact here)
  return (f >= -2147483648.0f && f < 2147483648.0f) ? i : (int)0x80000000;
}
__device__ __forceinline__ int cvt_x86_rt(float f, int rm) {
  return rm ? cvt_x86<1>(f) : cvt_x86<0>(f);
}

// vpmovusdb: unsigned saturation of the dword bit pattern
__device__ __forceinline__ unsigned sat_u8_bits(int v) {
  return min((unsigned)v, 255u);
}
// vpmovsdb: signed saturation
__device__ __forceinline__ int sat_s8(int v) { return max(-128, min(127, v)); }

// float(acc) (+ bias) * scale, optional ReLU; bias is already f32 (the host
// converts s8/u8/s32 bias with the same RNE int->f32 conversion vcvtdq2ps does;
// adding +0.0f for "no bias" is the identity because float(int) is never -0.0)
__device__ __forceinline__ float requant(int acc, float bias, float scale, bool relu) {
  float f = __int2float_rn(acc);
  f = __fadd_rn(f, bias);
  f = __fmul_rn(f, scale);
  return relu ? relu_x86(f) : f;
}

// 4 u8 x 4 s8 -> s32, exact (one dword lane of vpdpbusd)
__device__ __forceinline__ int dot4_u8s8(unsigned a, int w) {
  int s = 0;
  s += (int)(a & 0xffu) * (int)(int8_t)(w & 0xff);
  s += (int)((a >> 8) & 0xffu) * (int)(int8_t)((w >> 8) & 0xff);
  s += (int)((a >> 16) & 0xffu) * (int)(int8_t)((w >> 16) & 0xff);
  s += (int)(a >> 24) * (int)(w >> 24);
  return s;
}

// Kernel arguments common to both conv variants.  `consts` holds, as f32/s32
// arrays in natural channel order: comp0[oc] bias0[oc] scale0[oc] comp1[oc1]
// bias1[oc1] scale1[oc1]  (comp* = 128 * sum of the channel's weights as an exact
// f32, used by the MFMA variant's u8 -> s8 offset trick; zero for the generic variant).
struct ConvArgs {
  const uint8_t *src;
  void *dst;
  const int8_t *wei;    // generic: OIhw4i16o4i as given; mfma: packed fragments
  const int8_t *wei1;
  const float *consts;
  int bs, ic, ih, iw, oc, oh, ow, kh, kw, sh, sw, pt, pl, oc1;
  int dst_dt, relu0, relu1, rm0, rm1;
  int rows_per_unit, units_per_image;
};

// kernel arguments of concat.hip (shared with the host side in dfx_api.hip)
constexpr int CONCAT_MAX_INPUTS = 64;

struct ConcatArgs {
  const unsigned char *src[CONCAT_MAX_INPUTS];
  int chunk_end[CONCAT_MAX_INPUTS];  // exclusive prefix end, in 16-byte chunks per pixel
  int n_inputs;
  int chunks_per_px;  // of dst
  long long total_chunks;
  int dt, relu;
  unsigned char *dst;
};

struct PoolArgs {
  const unsigned char *src;
  unsigned char *dst;
  int bs, c, ih, iw, oh, ow, kh, kw, sh, sw, pad_t, pad_l, dt;
  int algo;             // dfx_pool_algo
  int vec;              // 1: one 16-byte channel group per thread (c * element size a multiple of 16)
  int groups;           // work items per output pixel
  long long total;      // work items
};

constexpr int ELTWISE_MAX_INPUTS = 8;
struct EltwiseArgs {
  const unsigned char *src[ELTWISE_MAX_INPUTS];
  unsigned char *dst;
  int n_inputs, dt, relu;
  long long elems;
};

}  // namespace dfx
