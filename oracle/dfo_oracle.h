/*
 * dfo_oracle.h -- CPU oracle for the deep-fusion hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the *intended* arithmetic of the reference's
 * fused int8 conv3x3+relu+conv1x1(+relu) primitive and of its concat(+relu)
 * primitive.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may link or call it; the product (deep-fusion_amd/) never does.
 *
 * PARITY STATUS
 *   conv / fused conv : "parity unpinned".  The reference holds no golden vectors
 *       or known-answer tests for this path (test/test_conv.cc:82 is an empty
 *       body) and cannot be built here (every src/ TU needs the un-vendored
 *       Xbyak, cmake/external/xbyak.cmake:31-32).  The restatement is pinned
 *       instead by (a) an independent float64/numpy formulation in
 *       tests/test_oracle.py and (b) dfo_avx512.c, which executes the very
 *       AVX-512 instructions the reference JIT emits (vpdpbusd, vcvtdq2ps,
 *       vaddps, vmulps, vmaxps, vcvtps2dq{rn,rd}, vpmovusdb, vpmovsdb).
 *   concat(+relu)     : pinned by the semantics test/test_concat.cc:31-87 checks
 *       against (MKL-DNN concat followed by eltwise_relu == plain channel concat
 *       then per-element max(0,x) in the element's own type).
 *
 * Reference files restated (all paths relative to /root/reference):
 *   3x3 MAC            src/jit_conv_kernel.cc:317-393  (compute_loop)
 *   requant stage 0    src/jit_conv_kernel.cc:218-305  (store_output)
 *   1x1 MAC            src/jit_conv_kernel.cc:143-191  (compute1x1_loop)
 *   requant stage 1    src/jit_conv_kernel.cc:50-141   (store_1x1output)
 *   weight layout      src/jit_conv_kernel.cc:329-338, :163-166
 *   padding            src/jit_conv_kernel.h:120-127, src/op_conv.cc:218-220
 *   shape rules        src/op_conv.cc:262-365, util/math_func.cc:22-24
 *   concat             src/jit_concat_kernel.cc:30-91, src/op_concat.cc:22-72
 * minus the defects listed in SURVEY.md section 8(a) ("Reference defects").
 */
#ifndef DFO_ORACLE_H
#define DFO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numeric values follow deepfusion::memory::dtype (include/deepfusion.h:66-72) */
enum { DFO_UNDEF = 0, DFO_F32 = 1, DFO_S32 = 2, DFO_S8 = 3, DFO_U8 = 4 };
/* deepfusion::round_mode (include/deepfusion.h:46-49) */
enum { DFO_ROUND_NEAREST = 0, DFO_ROUND_DOWN = 1 };

typedef struct dfo_conv_desc {
  int bs;             /* batch */
  int ic, ih, iw;     /* conv0 input  */
  int oc, oh, ow;     /* conv0 output */
  int kh, kw, sh, sw, pt, pl;
  int oc1x1;          /* 0: unfused conv; >0: fused 1x1 output channels */
  int dst_dt;         /* DFO_F32/S32/S8/U8 */
  int bia0_dt;        /* DFO_UNDEF = no bias */
  int bia1_dt;
  int relu0, relu1;   /* conv0_relu / conv1_relu flags as passed to conv() */
  int rm0, rm1;       /* round modes */
  int nscale0;        /* 1 or oc     */
  int nscale1;        /* 1 or oc1x1  */
} dfo_conv_desc;

/* byte offset of element (o,i,h,w) inside an OIhw4i16o4i s8 tensor with
 * I input channels (multiple of 16) and KHxKW taps:
 *   [o/16][i/16][kh][kw][(i%16)/4][o%16][i%4]   (jit_conv_kernel.cc:333-338) */
size_t dfo_blocked_wei_offset(int o, int i, int h, int w, int I, int KH, int KW);

/* plain oihw -> OIhw4i16o4i; O and I must be multiples of 16. */
void dfo_reorder_oihw_to_blocked(const int8_t *oihw, int8_t *blk, int O, int I,
                                 int KH, int KW);

/* Scalar reference. Buffers: src NHWC u8; wei OIhw4i16o4i s8 {oc,ic,kh,kw};
 * wei1 OIhw4i16o4i s8 {oc1x1,oc,1,1}; bias arrays typed by bia*_dt (may be
 * NULL when dt == DFO_UNDEF); dst NHWC of dst_dt.  Returns 0, or -1 when the
 * descriptor violates the reference's shape rules. */
int dfo_conv_scalar(const dfo_conv_desc *d, const uint8_t *src,
                    const int8_t *wei, const void *bia0, const float *scales0,
                    const int8_t *wei1, const void *bia1, const float *scales1,
                    void *dst);

/* Same contract, executed with the AVX-512(+VNNI) instructions the reference
 * JIT emits, OpenMP over (n, oh).  Returns -2 when the host lacks AVX-512 VNNI
 * (the caller then uses dfo_conv_scalar / dfo_conv_scalar_mt). */
int dfo_conv_avx512(const dfo_conv_desc *d, const uint8_t *src,
                    const int8_t *wei, const void *bia0, const float *scales0,
                    const int8_t *wei1, const void *bia1, const float *scales1,
                    void *dst);
int dfo_have_avx512_vnni(void);
int dfo_num_threads(void);
void dfo_set_num_threads(int n); /* OpenMP threads used by the *_mt / avx512 entry points */

/* scalar reference with the (n, oh) loop under OpenMP */
int dfo_conv_scalar_mt(const dfo_conv_desc *d, const uint8_t *src,
                       const int8_t *wei, const void *bia0,
                       const float *scales0, const int8_t *wei1,
                       const void *bia1, const float *scales1, void *dst);

/* dst[n,h,w, off_i + c] = relu?(src_i[n,h,w,c]); NHWC, dt in DFO_*.
 * Returns -1 when a channel count violates the reference's block rule
 * (jit_concat_kernel.cc:155-196: every C_i divisible by 16 for 1-byte types,
 * by 4 for 4-byte types). */
int dfo_concat(int n_inputs, const void *const *srcs, const int *channels,
               int bs, int h, int w, int dt, int post_relu, void *dst);

/* The reference's roadmap ops (README.md:64-65); semantics of the MKL-DNN pipeline its test builds
 * (test/test_conv_relu_pooling.cc:165-226 pooling_max: padding takes no part; :118-124 sum post-op).
 * No reference implementation exists: parity unpinned.  dfo_maxpool returns -2 when an output window
 * lies entirely in the padding. */
int dfo_maxpool(const void *src, void *dst, int bs, int c, int ih, int iw, int oh, int ow, int kh, int kw,
                int sh, int sw, int pad_t, int pad_l, int dt);
/* average pooling: algo 1 = divisor kh * kw (include padding), 2 = positions inside the input; integer types
 * exact sum, f32 division, nearest-even rounding, saturation (parity unpinned) */
int dfo_avgpool(const void *src, void *dst, int bs, int c, int ih, int iw, int oh, int ow, int kh, int kw,
                int sh, int sw, int pad_t, int pad_l, int dt, int algo);
/* dst = relu?(saturate(sum_k src_k)): integers exact + saturated to the dtype, f32 left to right */
int dfo_eltwise_sum(int n_inputs, const void *const *srcs, void *dst, long long elems, int dt, int post_relu);

/* conv_output_size, util/math_func.cc:22-24 */
int dfo_conv_out_size(int image, int kernel, int stride, int padding);

#ifdef __cplusplus
}
#endif
#endif
