/*
 * dfo_oracle.c -- scalar CPU oracle (TEST INFRASTRUCTURE ONLY, see dfo_oracle.h).
 *
 * Compile with -ffp-contract=off: the reference performs the bias add and the
 * scale multiply as two separately rounded f32 instructions (vaddps then vmulps,
 * /root/reference/src/jit_conv_kernel.cc:260-263 and :96-100); an FMA would
 * change integer outputs after rounding.
 */
#include "dfo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int dfo_conv_out_size(int image, int kernel, int stride, int padding) {
  /* util/math_func.cc:22-24 */
  return (image + 2 * padding - kernel) / stride + 1;
}

size_t dfo_blocked_wei_offset(int o, int i, int h, int w, int I, int KH, int KW) {
  /* jit_conv_kernel.cc:333-338 (kernel_offset) + :326 (kh shift) */
  size_t nb_ic = (size_t)I / 16;
  size_t off = (size_t)(o / 16) * nb_ic * KH * KW * 256;
  off += (size_t)(i / 16) * KH * KW * 256;
  off += ((size_t)h * KW + w) * 256;
  off += (size_t)((i % 16) / 4) * 64;
  off += (size_t)(o % 16) * 4;
  off += (size_t)(i % 4);
  return off;
}

void dfo_reorder_oihw_to_blocked(const int8_t *oihw, int8_t *blk, int O, int I,
                                 int KH, int KW) {
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int h = 0; h < KH; ++h)
        for (int w = 0; w < KW; ++w)
          blk[dfo_blocked_wei_offset(o, i, h, w, I, KH, KW)] =
              oihw[(((size_t)o * I + i) * KH + h) * KW + w];
}

/* ---- the f32 requantisation steps, one x86 instruction each ---- */

/* vcvtdq2ps of a bias element (jit_conv_kernel.cc:235-255, :68-86) */
static inline float bias_as_f32(const void *bia, int dt, int c) {
  switch (dt) {
    case DFO_F32: return ((const float *)bia)[c];
    case DFO_S32: return (float)((const int32_t *)bia)[c];
    case DFO_S8:  return (float)((const int8_t *)bia)[c];   /* vpmovsxbd */
    case DFO_U8:  return (float)((const uint8_t *)bia)[c];  /* vpmovzxbd */
    default:      return 0.0f;
  }
}

/* vmaxps(dst, zero, v): returns the second source unless zero > v
 * (so -0.0 and NaN pass through), jit_conv_kernel.cc:264-266, :102-104 */
static inline float relu_x86(float v) { return (0.0f > v) ? 0.0f : v; }

/* vcvtps2dq {rn-sae|rd-sae}: out-of-range and NaN give the integer indefinite
 * 0x80000000 (jit_conv_kernel.cc:267-274, :105-112) */
static inline int32_t cvt_x86(float f, int rm) {
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
  if (rm == DFO_ROUND_DOWN) return (int32_t)floorf(f);
  return (int32_t)nearbyintf(f); /* default FE_TONEAREST = ties-to-even */
}

/* vpmovusdb: unsigned saturation of the dword *bit pattern* */
static inline uint8_t sat_u8_from_bits(int32_t v) {
  return ((uint32_t)v > 255u) ? 255u : (uint8_t)v;
}
/* vpmovsdb: signed saturation */
static inline int8_t sat_s8(int32_t v) {
  return (int8_t)(v < -128 ? -128 : (v > 127 ? 127 : v));
}

static inline float requant_f32(int32_t acc, const void *bia, int bia_dt, int c,
                                const float *scales, int nscale, int relu) {
  float f = (float)acc;                           /* vcvtdq2ps */
  if (bia_dt != DFO_UNDEF) f = f + bias_as_f32(bia, bia_dt, c); /* vaddps */
  f = f * scales[nscale > 1 ? c : 0];             /* vmulps    */
  if (relu) f = relu_x86(f);                      /* vmaxps    */
  return f;
}

static inline void store_typed(void *dst, size_t idx, int dt, float f, int rm) {
  switch (dt) {
    case DFO_F32: ((float *)dst)[idx] = f; break;
    case DFO_S32: ((int32_t *)dst)[idx] = cvt_x86(f, rm); break;
    case DFO_S8:  ((int8_t *)dst)[idx] = sat_s8(cvt_x86(f, rm)); break;
    case DFO_U8:  ((uint8_t *)dst)[idx] = sat_u8_from_bits(cvt_x86(f, rm)); break;
    default: break;
  }
}

static int desc_ok(const dfo_conv_desc *d) {
  /* op_conv.cc:286-346 + jit_conv_kernel.cc:586-592, :619-621, :662-671 */
  if (d->bs <= 0 || d->ic <= 0 || d->oc <= 0) return 0;
  if (d->ic % 16 || d->oc % 16) return 0;
  if (d->oh != dfo_conv_out_size(d->ih, d->kh, d->sh, d->pt)) return 0;
  if (d->ow != dfo_conv_out_size(d->iw, d->kw, d->sw, d->pl)) return 0;
  if (d->oh <= 0 || d->ow <= 0) return 0;
  if (d->dst_dt < DFO_F32 || d->dst_dt > DFO_U8) return 0;
  if (d->nscale0 != 1 && d->nscale0 != d->oc) return 0;
  if (d->oc1x1) {
    if (d->oc1x1 % 16) return 0;
    if (d->nscale1 != 1 && d->nscale1 != d->oc1x1) return 0;
  }
  return 1;
}

/* one output pixel: exact s32 conv0 accumulators for all oc into acc0[].
 * Walks the blocked weight tensor in its storage order
 * [oc/16][ic/16][kh][kw][4i][16o][4i] (jit_conv_kernel.cc:333-338). */
static void conv0_pixel(const dfo_conv_desc *d, const uint8_t *src_n,
                        const int8_t *wei, int oy, int ox, int32_t *acc0) {
  const int nb_ic = d->ic / 16, nb_oc = d->oc / 16;
  for (int oc = 0; oc < d->oc; ++oc) acc0[oc] = 0;
  for (int kh = 0; kh < d->kh; ++kh) {
    int iy = oy * d->sh - d->pt + kh;
    if (iy < 0 || iy >= d->ih) continue; /* kh_padding, op_conv.cc:218-220 */
    for (int kw = 0; kw < d->kw; ++kw) {
      int ix = ox * d->sw - d->pl + kw;
      if (ix < 0 || ix >= d->iw) continue; /* get_ow_start/end, jit_conv_kernel.h:120-127 */
      const uint8_t *px = src_n + ((size_t)iy * d->iw + ix) * d->ic;
      for (int ocb = 0; ocb < nb_oc; ++ocb)
        for (int icb = 0; icb < nb_ic; ++icb) {
          const int8_t *blk =
              wei + (((size_t)ocb * nb_ic + icb) * d->kh * d->kw + (size_t)kh * d->kw + kw) * 256;
          for (int i4 = 0; i4 < 4; ++i4)
            for (int o = 0; o < 16; ++o) {
              /* one dword lane of vpdpbusd: 4 u8 x 4 s8, exact */
              int32_t s = 0;
              for (int i = 0; i < 4; ++i)
                s += (int32_t)px[icb * 16 + i4 * 4 + i] * (int32_t)blk[i4 * 64 + o * 4 + i];
              acc0[ocb * 16 + o] += s;
            }
        }
    }
  }
}

static void conv_rows(const dfo_conv_desc *d, const uint8_t *src,
                      const int8_t *wei, const void *bia0, const float *scales0,
                      const int8_t *wei1, const void *bia1, const float *scales1,
                      void *dst, long row_begin, long row_end) {
  int32_t *acc0 = (int32_t *)malloc(sizeof(int32_t) * (size_t)d->oc);
  uint8_t *mid = (uint8_t *)malloc((size_t)d->oc);
  const int fused = d->oc1x1 > 0;
  for (long row = row_begin; row < row_end; ++row) {
    int n = (int)(row / d->oh), oy = (int)(row % d->oh);
    const uint8_t *src_n = src + (size_t)n * d->ih * d->iw * d->ic;
    for (int ox = 0; ox < d->ow; ++ox) {
      conv0_pixel(d, src_n, wei, oy, ox, acc0);
      size_t pix = ((size_t)n * d->oh + oy) * d->ow + ox;
      if (!fused) {
        /* store_output, unfused branch (jit_conv_kernel.cc:279-297) */
        int relu = d->relu0 || d->dst_dt == DFO_U8;
        for (int oc = 0; oc < d->oc; ++oc) {
          float f = requant_f32(acc0[oc], bia0, d->bia0_dt, oc, scales0, d->nscale0, relu);
          store_typed(dst, pix * d->oc + oc, d->dst_dt, f, d->rm0);
        }
        continue;
      }
      /* fused: ReLU always, always narrowed to u8 (jit_conv_kernel.cc:264-277) */
      for (int oc = 0; oc < d->oc; ++oc) {
        float f = requant_f32(acc0[oc], bia0, d->bia0_dt, oc, scales0, d->nscale0, 1);
        mid[oc] = sat_u8_from_bits(cvt_x86(f, d->rm0));
      }
      int relu1 = d->relu1 || d->dst_dt == DFO_U8;
      for (int o1 = 0; o1 < d->oc1x1; ++o1) {
        int32_t acc1 = 0;
        for (int oc = 0; oc < d->oc; ++oc)
          acc1 += (int32_t)mid[oc] *
                  (int32_t)wei1[dfo_blocked_wei_offset(o1, oc, 0, 0, d->oc, 1, 1)];
        float f = requant_f32(acc1, bia1, d->bia1_dt, o1, scales1, d->nscale1, relu1);
        store_typed(dst, pix * d->oc1x1 + o1, d->dst_dt, f, d->rm1);
      }
    }
  }
  free(acc0);
  free(mid);
}

int dfo_conv_scalar(const dfo_conv_desc *d, const uint8_t *src, const int8_t *wei,
                    const void *bia0, const float *scales0, const int8_t *wei1,
                    const void *bia1, const float *scales1, void *dst) {
  if (!desc_ok(d)) return -1;
  conv_rows(d, src, wei, bia0, scales0, wei1, bia1, scales1, dst, 0,
            (long)d->bs * d->oh);
  return 0;
}

int dfo_conv_scalar_mt(const dfo_conv_desc *d, const uint8_t *src,
                       const int8_t *wei, const void *bia0, const float *scales0,
                       const int8_t *wei1, const void *bia1, const float *scales1,
                       void *dst) {
  if (!desc_ok(d)) return -1;
  long rows = (long)d->bs * d->oh;
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r)
    conv_rows(d, src, wei, bia0, scales0, wei1, bia1, scales1, dst, r, r + 1);
  return 0;
}

void dfo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int dfo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int dfo_concat(int n_inputs, const void *const *srcs, const int *channels, int bs,
               int h, int w, int dt, int post_relu, void *dst) {
  size_t esz = (dt == DFO_F32 || dt == DFO_S32) ? 4 : 1;
  int blk = (esz == 1) ? 16 : 4; /* smallest block tried, jit_concat_kernel.cc:155-176 */
  int oc = 0;
  for (int i = 0; i < n_inputs; ++i) {
    if (channels[i] <= 0 || channels[i] % blk) return -1;
    oc += channels[i];
  }
  size_t npix = (size_t)bs * h * w;
  for (size_t p = 0; p < npix; ++p) {
    size_t off = 0;
    for (int i = 0; i < n_inputs; ++i) {
      size_t c = (size_t)channels[i];
      for (size_t k = 0; k < c; ++k) {
        size_t si = p * c + k, di = p * (size_t)oc + off + k;
        switch (dt) {
          case DFO_F32: {
            float v = ((const float *)srcs[i])[si];
            ((float *)dst)[di] = post_relu ? relu_x86(v) : v; /* vmaxps(zero, v) */
          } break;
          case DFO_S32: {
            int32_t v = ((const int32_t *)srcs[i])[si];
            ((int32_t *)dst)[di] = (post_relu && v < 0) ? 0 : v;
          } break;
          case DFO_S8: {
            int8_t v = ((const int8_t *)srcs[i])[si];
            ((int8_t *)dst)[di] = (post_relu && v < 0) ? 0 : v;
          } break;
          default: /* u8: ReLU is the identity on unsigned data */
            ((uint8_t *)dst)[di] = ((const uint8_t *)srcs[i])[si];
        }
      }
      off += c;
    }
  }
  return 0;
}

/* ---- the reference's roadmap ops (README.md:64-65), semantics of the MKL-DNN pipeline in
 * test/test_conv_relu_pooling.cc:165-226 (pooling_max over NHWC, padding takes no part) and
 * :118-124 (sum post-op).  The reference has no implementation: parity unpinned. ---- */
int dfo_maxpool(const void *src, void *dst, int bs, int c, int ih, int iw, int oh, int ow, int kh, int kw,
                int sh, int sw, int pad_t, int pad_l, int dt) {
  if (!src || !dst || dt < DFO_F32 || dt > DFO_U8) return -1;
  for (int n = 0; n < bs; ++n)
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox)
        for (int ch = 0; ch < c; ++ch) {
          int first = 1;
          float bf = 0.f;
          long long bi = 0;
          for (int ky = 0; ky < kh; ++ky) {
            const int y = oy * sh - pad_t + ky;
            if (y < 0 || y >= ih) continue;
            for (int kx = 0; kx < kw; ++kx) {
              const int x = ox * sw - pad_l + kx;
              if (x < 0 || x >= iw) continue;
              const size_t i = (((size_t)n * ih + y) * iw + x) * c + ch;
              if (dt == DFO_F32) {
                const float v = ((const float *)src)[i];
                bf = first ? (-INFINITY > v ? -INFINITY : v) : (bf > v ? bf : v); /* vmaxps(acc, v) order */
              } else {
                const long long v = dt == DFO_S32 ? ((const int32_t *)src)[i]
                                  : dt == DFO_S8 ? ((const int8_t *)src)[i] : ((const uint8_t *)src)[i];
                bi = first ? v : (bi > v ? bi : v);
              }
              first = 0;
            }
          }
          if (first) return -2; /* a window entirely in the padding */
          const size_t o = (((size_t)n * oh + oy) * ow + ox) * c + ch;
          if (dt == DFO_F32) ((float *)dst)[o] = bf;
          else if (dt == DFO_S32) ((int32_t *)dst)[o] = (int32_t)bi;
          else if (dt == DFO_S8) ((int8_t *)dst)[o] = (int8_t)bi;
          else ((uint8_t *)dst)[o] = (uint8_t)bi;
        }
  return 0;
}

/* algo 1: average including padding (divisor kh * kw), 2: excluding it (divisor = positions inside the input);
 * integer types: exact sum, (float)sum / (float)count, nearbyintf (nearest even), saturation; f32: sum in
 * window order, one division (MKL-DNN reference pooling arithmetic; parity unpinned) */
int dfo_avgpool(const void *src, void *dst, int bs, int c, int ih, int iw, int oh, int ow, int kh, int kw,
                int sh, int sw, int pad_t, int pad_l, int dt, int algo) {
  if (!src || !dst || dt < DFO_F32 || dt > DFO_U8 || (algo != 1 && algo != 2)) return -1;
  for (int n = 0; n < bs; ++n)
    for (int oy = 0; oy < oh; ++oy)
      for (int ox = 0; ox < ow; ++ox)
        for (int ch = 0; ch < c; ++ch) {
          float sf = 0.f;
          long long si = 0;
          int inside = 0;
          for (int ky = 0; ky < kh; ++ky) {
            const int y = oy * sh - pad_t + ky;
            if (y < 0 || y >= ih) continue;
            for (int kx = 0; kx < kw; ++kx) {
              const int x = ox * sw - pad_l + kx;
              if (x < 0 || x >= iw) continue;
              const size_t i = (((size_t)n * ih + y) * iw + x) * c + ch;
              if (dt == DFO_F32) sf = sf + ((const float *)src)[i];
              else si += dt == DFO_S32 ? ((const int32_t *)src)[i] : dt == DFO_S8 ? ((const int8_t *)src)[i]
                                                                              : ((const uint8_t *)src)[i];
              ++inside;
            }
          }
          if (!inside) return -2;
          const int count = algo == 1 ? kh * kw : inside;
          const size_t o = (((size_t)n * oh + oy) * ow + ox) * c + ch;
          if (dt == DFO_F32) {
            ((float *)dst)[o] = sf / (float)count;
          } else {
            const float q = nearbyintf((float)si / (float)count);
            if (dt == DFO_S32) ((int32_t *)dst)[o] = q >= 2147483648.0f ? INT32_MAX : q <= -2147483648.0f ? INT32_MIN : (int32_t)q;
            else if (dt == DFO_S8) ((int8_t *)dst)[o] = (int8_t)(q > 127.f ? 127 : q < -128.f ? -128 : (int)q);
            else ((uint8_t *)dst)[o] = (uint8_t)(q > 255.f ? 255 : q < 0.f ? 0 : (int)q);
          }
        }
  return 0;
}

int dfo_eltwise_sum(int n_inputs, const void *const *srcs, void *dst, long long elems, int dt, int post_relu) {
  if (!srcs || !dst || n_inputs < 1 || dt < DFO_F32 || dt > DFO_U8) return -1;
  for (long long i = 0; i < elems; ++i) {
    if (dt == DFO_F32) {
      float a = ((const float *)srcs[0])[i];
      for (int k = 1; k < n_inputs; ++k) a = a + ((const float *)srcs[k])[i];
      if (post_relu) a = (0.0f > a) ? 0.0f : a; /* vmaxps(zero, a) */
      ((float *)dst)[i] = a;
    } else {
      long long a = 0;
      for (int k = 0; k < n_inputs; ++k)
        a += dt == DFO_S32 ? ((const int32_t *)srcs[k])[i] : dt == DFO_S8 ? ((const int8_t *)srcs[k])[i]
                                                                         : ((const uint8_t *)srcs[k])[i];
      if (post_relu && a < 0) a = 0;
      const long long lo = dt == DFO_S32 ? INT32_MIN : dt == DFO_S8 ? -128 : 0;
      const long long hi = dt == DFO_S32 ? INT32_MAX : dt == DFO_S8 ? 127 : 255;
      a = a < lo ? lo : (a > hi ? hi : a);
      if (dt == DFO_S32) ((int32_t *)dst)[i] = (int32_t)a;
      else if (dt == DFO_S8) ((int8_t *)dst)[i] = (int8_t)a;
      else ((uint8_t *)dst)[i] = (uint8_t)a;
    }
  }
  return 0;
}
