/*
 * dfo_avx512.c -- AVX-512(+VNNI) witness of the oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Executes, through intrinsics, the instruction sequence the reference's Xbyak
 * JIT emits for one register tile, so the integer and f32 behaviour is that of
 * the very x86 instructions (not a C model of them):
 *   compute_loop      jit_conv_kernel.cc:358-389  vpbroadcastd / vmovups / vpdpbusd
 *   store_output      jit_conv_kernel.cc:256-277  vcvtdq2ps, vaddps, vmulps, vmaxps,
 *                                                 vcvtps2dq{rn,rd}-sae, vpmovusdb/vpmovsdb
 *   compute1x1_loop   jit_conv_kernel.cc:161-190
 *   store_1x1output   jit_conv_kernel.cc:88-130
 * with the driver defects of SURVEY.md 8(a) corrected (NHWC row strides, x256
 * weight strides, scale broadcast when count == 1, the intermediate always
 * converted to integer before vpmovusdb).  The register tile follows the
 * reference blocking (ur_w = 28 / (nb_oc_blocking + 1), jit_conv_kernel.cc:653)
 * and the work split follows op_conv.cc:149-156 (contiguous (n, oh) ranges per
 * OpenMP thread).  It is also the timed CPU baseline in bench.py ("port").
 *
 * Built with -mavx512f -mavx512bw -mavx512vl -mavx512vnni; every entry point
 * checks CPUID first and returns -2 when the host cannot run it.
 */
#include "dfo_oracle.h"

#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int dfo_have_avx512_vnni(void) {
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
         __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vnni");
}

typedef struct {
  const dfo_conv_desc *d;
  const uint8_t *src;
  const int8_t *wei, *wei1;
  const void *bia0, *bia1;
  const float *sc0, *sc1;
  void *dst;
  int nb_ic, nb_oc, nb_oc1;
} ctx_t;

#define MAX_UR 14
#define MAX_NB 4

/* 16 bias lanes as f32: vmovups | vpmovsxbd | vpmovzxbd, then vcvtdq2ps */
static inline __m512 load_bias16(const void *bia, int dt, int c0) {
  switch (dt) {
    case DFO_F32: return _mm512_loadu_ps((const float *)bia + c0);
    case DFO_S32: return _mm512_cvtepi32_ps(_mm512_loadu_si512((const int32_t *)bia + c0));
    case DFO_S8:
      return _mm512_cvtepi32_ps(
          _mm512_cvtepi8_epi32(_mm_loadu_si128((const __m128i *)((const int8_t *)bia + c0))));
    case DFO_U8:
      return _mm512_cvtepi32_ps(
          _mm512_cvtepu8_epi32(_mm_loadu_si128((const __m128i *)((const uint8_t *)bia + c0))));
    default: return _mm512_setzero_ps();
  }
}

static inline __m512 load_scale16(const float *s, int n, int c0) {
  return n > 1 ? _mm512_loadu_ps(s + c0) : _mm512_set1_ps(s[0]);
}

static inline __m512i cvt_ps_epi32(__m512 f, int rm) {
  return rm == DFO_ROUND_DOWN
             ? _mm512_cvt_roundps_epi32(f, _MM_FROUND_TO_NEG_INF | _MM_FROUND_NO_EXC)
             : _mm512_cvt_roundps_epi32(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
}

static inline void store16(void *dst, size_t idx, int dt, __m512 f, int rm) {
  switch (dt) {
    case DFO_F32: _mm512_storeu_ps((float *)dst + idx, f); break;
    case DFO_S32: _mm512_storeu_si512((int32_t *)dst + idx, cvt_ps_epi32(f, rm)); break;
    case DFO_S8:
      _mm_storeu_si128((__m128i *)((int8_t *)dst + idx),
                       _mm512_cvtsepi32_epi8(cvt_ps_epi32(f, rm))); /* vpmovsdb */
      break;
    default:
      _mm_storeu_si128((__m128i *)((uint8_t *)dst + idx),
                       _mm512_cvtusepi32_epi8(cvt_ps_epi32(f, rm))); /* vpmovusdb */
  }
}

/* One register tile: UR output pixels [ox0, ox0+UR) of row (n, oy), conv0 output
 * channel blocks [ocb0, ocb0+NB).  Writes either the typed unfused output or the
 * u8 intermediate mid[j][oc]. */
static inline __attribute__((always_inline)) void conv0_tile(
    const ctx_t *c, int n, int oy, int ox0, int ocb0, const int NB, const int UR,
    uint8_t *mid) {
  const dfo_conv_desc *d = c->d;
  __m512i acc[MAX_NB][MAX_UR];
#pragma GCC unroll 4
  for (int k = 0; k < NB; ++k)
#pragma GCC unroll 14
    for (int j = 0; j < UR; ++j) acc[k][j] = _mm512_setzero_si512();

  const uint8_t *src_n = c->src + (size_t)n * d->ih * d->iw * d->ic;
  for (int kh = 0; kh < d->kh; ++kh) {
    int iy = oy * d->sh - d->pt + kh;
    if (iy < 0 || iy >= d->ih) continue;
    for (int kw = 0; kw < d->kw; ++kw) {
      /* valid pixel range of this tap inside the tile (get_ow_start/end) */
      int j0 = 0, j1 = UR;
      while (j0 < j1 && (ox0 + j0) * d->sw - d->pl + kw < 0) ++j0;
      while (j1 > j0 && (ox0 + j1 - 1) * d->sw - d->pl + kw >= d->iw) --j1;
      if (j0 >= j1) continue;
      for (int cc = 0; cc < c->nb_ic; ++cc) {
        for (int i4 = 0; i4 < 4; ++i4) {
          __m512i inp[MAX_UR];
#pragma GCC unroll 14
          for (int j = 0; j < UR; ++j) {
            if (j < j0 || j >= j1) continue;
            int ix = (ox0 + j) * d->sw - d->pl + kw;
            int32_t v;
            memcpy(&v, src_n + ((size_t)iy * d->iw + ix) * d->ic + cc * 16 + i4 * 4, 4);
            inp[j] = _mm512_set1_epi32(v); /* vpbroadcastd */
          }
#pragma GCC unroll 4
          for (int k = 0; k < NB; ++k) {
            const int8_t *wp = c->wei +
                               (((size_t)(ocb0 + k) * c->nb_ic + cc) * d->kh * d->kw +
                                (size_t)kh * d->kw + kw) * 256 + i4 * 64;
            __m512i w = _mm512_loadu_si512(wp);
#pragma GCC unroll 14
            for (int j = 0; j < UR; ++j) {
              if (j < j0 || j >= j1) continue;
              acc[k][j] = _mm512_dpbusd_epi32(acc[k][j], inp[j], w); /* vpdpbusd */
            }
          }
        }
      }
    }
  }

  const int fused = d->oc1x1 > 0;
  const int relu = fused || d->relu0 || d->dst_dt == DFO_U8;
  const __m512 zero = _mm512_setzero_ps();
#pragma GCC unroll 4
  for (int k = 0; k < NB; ++k) {
    int c0 = (ocb0 + k) * 16;
    __m512 bias = load_bias16(c->bia0, d->bia0_dt, c0);
    __m512 scale = load_scale16(c->sc0, d->nscale0, c0);
#pragma GCC unroll 14
    for (int j = 0; j < UR; ++j) {
      __m512 f = _mm512_cvtepi32_ps(acc[k][j]);
      if (d->bia0_dt != DFO_UNDEF) f = _mm512_add_ps(f, bias);
      f = _mm512_mul_ps(f, scale);
      if (relu) f = _mm512_max_ps(zero, f);
      if (fused) {
        __m128i u = _mm512_cvtusepi32_epi8(cvt_ps_epi32(f, d->rm0)); /* vpmovusdb */
        _mm_storeu_si128((__m128i *)(mid + (size_t)j * d->oc + c0), u);
      } else {
        size_t pix = ((size_t)n * d->oh + oy) * d->ow + ox0 + j;
        store16(c->dst, pix * d->oc + c0, d->dst_dt, f, d->rm0);
      }
    }
  }
}

static inline __attribute__((always_inline)) void conv1_tile(const ctx_t *c, int n, int oy,
                                                             int ox0, const int UR,
                                                             const uint8_t *mid) {
  const dfo_conv_desc *d = c->d;
  const int relu1 = d->relu1 || d->dst_dt == DFO_U8;
  const __m512 zero = _mm512_setzero_ps();
  for (int ob = 0; ob < c->nb_oc1; ++ob) {
    __m512i acc[MAX_UR];
#pragma GCC unroll 14
    for (int j = 0; j < UR; ++j) acc[j] = _mm512_setzero_si512();
    for (int k = 0; k < c->nb_oc; ++k)
      for (int i4 = 0; i4 < 4; ++i4) {
        __m512i w = _mm512_loadu_si512(c->wei1 + ((size_t)ob * c->nb_oc + k) * 256 + i4 * 64);
#pragma GCC unroll 14
        for (int j = 0; j < UR; ++j) {
          int32_t v; /* vmovd / vpextrd of the u8 intermediate, then vpbroadcastd */
          memcpy(&v, mid + (size_t)j * d->oc + k * 16 + i4 * 4, 4);
          acc[j] = _mm512_dpbusd_epi32(acc[j], _mm512_set1_epi32(v), w);
        }
      }
    __m512 bias = load_bias16(c->bia1, d->bia1_dt, ob * 16);
    __m512 scale = load_scale16(c->sc1, d->nscale1, ob * 16);
#pragma GCC unroll 14
    for (int j = 0; j < UR; ++j) {
      __m512 f = _mm512_cvtepi32_ps(acc[j]);
      if (d->bia1_dt != DFO_UNDEF) f = _mm512_add_ps(f, bias);
      f = _mm512_mul_ps(f, scale);
      if (relu1) f = _mm512_max_ps(zero, f);
      size_t pix = ((size_t)n * d->oh + oy) * d->ow + ox0 + j;
      store16(c->dst, pix * d->oc1x1 + ob * 16, d->dst_dt, f, d->rm1);
    }
  }
}

/* a full register tile of the row: every oc chunk, then the fused 1x1 */
static inline __attribute__((always_inline)) void row_tile(const ctx_t *c, int n, int oy,
                                                           int ox0, const int NB,
                                                           const int UR, uint8_t *mid) {
  for (int ocb0 = 0; ocb0 < c->nb_oc; ocb0 += NB) conv0_tile(c, n, oy, ox0, ocb0, NB, UR, mid);
  if (c->d->oc1x1 > 0) conv1_tile(c, n, oy, ox0, UR, mid);
}

#define DEF_TILE(NB, UR)                                                                  \
  static __attribute__((noinline)) void tile_##NB##_##UR(const ctx_t *c, int n, int oy,  \
                                                         int ox0, uint8_t *mid) {         \
    row_tile(c, n, oy, ox0, NB, UR, mid);                                                 \
  }
DEF_TILE(1, 14)
DEF_TILE(2, 9)
DEF_TILE(3, 7)
DEF_TILE(4, 5)

/* tails and unusual blockings: same arithmetic, bounds not known at compile time */
static __attribute__((noinline)) void tile_any(const ctx_t *c, int n, int oy, int ox0, int nb,
                                               int ur, uint8_t *mid) {
  row_tile(c, n, oy, ox0, nb, ur, mid);
}

static void run_row(const ctx_t *c, int n, int oy, int nb, int ur, uint8_t *mid) {
  const dfo_conv_desc *d = c->d;
  int ox = 0;
  for (; ox + ur <= d->ow; ox += ur) {
    if (nb == 4 && ur == 5) tile_4_5(c, n, oy, ox, mid);
    else if (nb == 2 && ur == 9) tile_2_9(c, n, oy, ox, mid);
    else if (nb == 3 && ur == 7) tile_3_7(c, n, oy, ox, mid);
    else if (nb == 1 && ur == 14) tile_1_14(c, n, oy, ox, mid);
    else tile_any(c, n, oy, ox, nb, ur, mid);
  }
  if (ox < d->ow) tile_any(c, n, oy, ox, nb, d->ow - ox, mid);
}

int dfo_conv_avx512(const dfo_conv_desc *d, const uint8_t *src, const int8_t *wei,
                    const void *bia0, const float *scales0, const int8_t *wei1,
                    const void *bia1, const float *scales1, void *dst) {
  if (!dfo_have_avx512_vnni()) return -2;
  if (d->ic % 16 || d->oc % 16 || (d->oc1x1 % 16)) return -1;
  if (d->oh != dfo_conv_out_size(d->ih, d->kh, d->sh, d->pt) ||
      d->ow != dfo_conv_out_size(d->iw, d->kw, d->sw, d->pl))
    return -1;
  ctx_t c = {d, src, wei, wei1, bia0, bia1, scales0, scales1, dst,
             d->ic / 16, d->oc / 16, d->oc1x1 / 16};
  /* blocking, jit_conv_kernel.cc:647-655 */
  int nb = c.nb_oc > 4 ? 4 : c.nb_oc;
  while (c.nb_oc % nb) --nb;
  int ur = 28 / (nb + 1);
  if (d->ow < ur) ur = d->ow;
  long rows = (long)d->bs * d->oh;
#pragma omp parallel
  {
    uint8_t *mid = (uint8_t *)malloc((size_t)MAX_UR * d->oc + 64);
#pragma omp for schedule(static)
    for (long r = 0; r < rows; ++r) run_row(&c, (int)(r / d->oh), (int)(r % d->oh), nb, ur, mid);
    free(mid);
  }
  return 0;
}
