// conv_mfma.cuh -- fused u8 x s8 conv3x3 (stride 1) + ReLU + requant + conv1x1
// (+ReLU) + requant as two chained int8-MFMA implicit GEMMs (gfx950 / CDNA4).
//
// Replaces the reference's JIT micro-kernel and its driver loops:
//   compute_loop / store_output        /root/reference/src/jit_conv_kernel.cc:317-393, :218-305
//   compute1x1_loop / store_1x1output  src/jit_conv_kernel.cc:143-191, :50-141
//   infer_conv0conv1                   src/op_conv.cc:140-260
//
// Design (one workgroup = one "unit" = TH output rows x TW output columns of
// one image; 4 waves; each wave walks 32-pixel tiles of the unit):
//
//  * conv0 is D0[oc][px] = sum_k W0[oc][k] * X[k][px] with
//    v_mfma_i32_32x32x32_i8: the packed s8 weights are the A operand (rows =
//    oc), the input pixels are the B operand (columns = px).  One MFMA eats 32
//    input channels of one (kh,kw) tap.  Both operands come from LDS with
//    ds_read_b128: the weight image is lane-linear (packed on the host), the
//    input halo tile is stored [row][col][ic] with a 16-byte-chunk XOR swizzle so
//    that the 64 B/pixel stride is bank-conflict free.
//  * MFMA i8 is signed x signed.  Activations are stored in LDS as (u8 xor 0x80)
//    = u8 - 128; the accumulators start at comp0[oc] = 128 * sum_k W0[oc][k], so
//    the s32 result is exact.  Zero padding is the byte 0x80 (= real 0), which
//    keeps one compensation constant valid at the borders.
//  * After conv0 a lane holds, for its pixel, 16 accumulators per 32-oc block at
//    oc = 32r + 8q + 4h + i (h = lane>>5).  They are requantised in registers
//    (ReLU, scale, round, saturate to u8) and packed 4 per dword; those 16 bytes
//    per block ARE the A-operand fragment of the 1x1 MFMA, because the 1x1
//    weights were packed on the host in exactly this k order.  The intermediate
//    activation never leaves the register file.
//  * conv1 is D1[px][oc1] = sum_oc mid[px][oc] * W1[oc][oc1]: lane = output
//    channel, registers = pixels, so bias/scale are per-lane constants.  The host
//    also permutes which channel each MFMA column computes: in a group of G
//    column blocks lane L owns channels 32G*cg + G*L + {0..G-1}, so every pixel
//    is written with one G*4-byte (s32/f32) or G-byte (s8/u8) store per lane
//    and a half-wave writes 128*G (or 32*G) contiguous bytes: full HBM lines.
//
// Supported here: kh = kw = 3, stride 1, pad in {0,1}, ic/oc/oc1x1 multiples of
// 32 (ic, oc <= 64 per instantiation list in conv_mfma_inst.inc).  Everything
// else goes to conv_generic.hip.
#pragma once

#include "dfx_device.cuh"

namespace dfx {

constexpr int MFMA_THREADS = 256;

__device__ __forceinline__ v16i mfma_i8(v4i a, v4i b, v16i c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

template <int CP>  // 16-byte chunks per pixel; chunk j of LDS pixel P sits at j ^ swz(P)
__device__ __forceinline__ int chunk_swizzle(int P) {
  if (CP == 2) return (P >> 3) & 1;
  if (CP == 4) return (P >> 2) & 3;
  return (P >> 1) & 7;  // CP == 8
}

// XCD-aware block remap (bijective form): blocks that share blockIdx%8 share an
// XCD/L2; give each XCD a contiguous run of units so vertically adjacent units
// (which share halo rows) hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int DST> struct DstT;
template <> struct DstT<DFX_F32> { typedef float type; };
template <> struct DstT<DFX_S32> { typedef int type; };
template <> struct DstT<DFX_S8> { typedef int8_t type; };
template <> struct DstT<DFX_U8> { typedef uint8_t type; };

// one pixel's G consecutive channels -> one store
template <int DST, int G>
__device__ __forceinline__ void store_group(void *dst, size_t elem, const float (&f)[G], int rm) {
  if (DST == DFX_F32) {
    float *p = reinterpret_cast<float *>(dst) + elem;
    if (G == 4) *reinterpret_cast<v4f *>(p) = v4f{f[0], f[1], f[2], f[3]};
    else if (G == 2) *reinterpret_cast<float2 *>(p) = float2{f[0], f[1]};
    else p[0] = f[0];
  } else if (DST == DFX_S32) {
    int v[G];
#pragma unroll
    for (int c = 0; c < G; ++c) v[c] = cvt_x86_rt(f[c], rm);
    int *p = reinterpret_cast<int *>(dst) + elem;
    if (G == 4) *reinterpret_cast<v4i *>(p) = v4i{v[0], v[1], v[2], v[3]};
    else if (G == 2) *reinterpret_cast<int2 *>(p) = int2{v[0], v[1]};
    else p[0] = v[0];
  } else {
    unsigned pk = 0;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      const int v = cvt_x86_rt(f[c], rm);
      const unsigned b = (DST == DFX_U8) ? sat_u8_bits(v) : ((unsigned)sat_s8(v) & 0xffu);
      pk |= b << (8 * c);
    }
    uint8_t *p = reinterpret_cast<uint8_t *>(dst) + elem;
    if (G == 4) *reinterpret_cast<unsigned *>(p) = pk;
    else if (G == 2) *reinterpret_cast<unsigned short *>(p) = (unsigned short)pk;
    else p[0] = (uint8_t)pk;
  }
}

struct MfmaGeom {  // unit decomposition chosen by the host (dfx_api.cpp)
  int th, tw;      // unit size in output rows / columns
  int uy, ux;      // units per image along y / x
  int linear;      // 1: tw == ow, pixels of a unit are numbered linearly across rows
                   // 0: tw % 32 == 0, every 32-pixel tile lies inside one row
};

template <int ICB, int OCB, int G, int DST>
__global__ __launch_bounds__(MFMA_THREADS, 2) void conv_mfma_fused_kernel(ConvArgs a, MfmaGeom g) {
  constexpr int IC = 32 * ICB, OC = 32 * OCB, CP = IC / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int OC1 = a.oc1, NCB = OC1 >> 5, NCG = NCB / G;
  unsigned char *w0s = smem;                                   // [OCB][9][ICB][64 lanes][16 B]
  unsigned char *w1s = w0s + OCB * 9 * ICB * 1024;             // [NCB][OCB][64 lanes][16 B]
  float *cst = reinterpret_cast<float *>(w1s + NCB * OCB * 1024);
  const int cst_bytes = (3 * (OC + OC1) * 4 + 15) & ~15;
  unsigned char *ins = reinterpret_cast<unsigned char *>(cst) + cst_bytes;  // halo tile

  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int upi = g.uy * g.ux;
  const int n = bid / upi, u = bid - n * upi;
  const int uyi = u / g.ux, uxi = u - uyi * g.ux;
  const int y0 = uyi * g.th, x0 = uxi * g.tw;
  const int th = min(g.th, a.oh - y0), tw = min(g.tw, a.ow - x0);

  // ---- stage weights + constants (already in fragment order: linear copy) ----
  {
    const v4i *s = reinterpret_cast<const v4i *>(a.wei);
    v4i *d = reinterpret_cast<v4i *>(w0s);
    for (int i = tid; i < OCB * 9 * ICB * 64; i += MFMA_THREADS) d[i] = s[i];
    s = reinterpret_cast<const v4i *>(a.wei1);
    d = reinterpret_cast<v4i *>(w1s);
    for (int i = tid; i < NCB * OCB * 64; i += MFMA_THREADS) d[i] = s[i];
    for (int i = tid; i < 3 * (OC + OC1); i += MFMA_THREADS) cst[i] = a.consts[i];
  }
  // ---- stage the input halo tile: rows y0-pt .. +th+1, cols x0-pl .. +tw+1 ----
  const int LW = tw + 2;
  {
    const uint8_t *src_n = a.src + (size_t)n * a.ih * a.iw * IC;
    const v4i pad = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    for (int lr = 0; lr < th + 2; ++lr) {
      const int iy = y0 - a.pt + lr;
      const bool rowok = iy >= 0 && iy < a.ih;
      for (int c = tid; c < LW * CP; c += MFMA_THREADS) {
        const int lc = c / CP, j = c % CP;
        const int ix = x0 - a.pl + lc;
        v4i v = pad;
        if (rowok && ix >= 0 && ix < a.iw)
          v = *reinterpret_cast<const v4i *>(src_n + ((size_t)iy * a.iw + ix) * IC + 16 * j) ^ pad;
        const int P = lr * LW + lc;
        *reinterpret_cast<v4i *>(ins + P * IC + 16 * (j ^ chunk_swizzle<CP>(P))) = v;
      }
    }
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int *comp0 = reinterpret_cast<const int *>(cst);
  const float *bias0 = cst + OC, *scale0 = cst + 2 * OC;
  const int *comp1 = reinterpret_cast<const int *>(cst + 3 * OC);
  const float *bias1 = cst + 3 * OC + OC1, *scale1 = cst + 3 * OC + 2 * OC1;
  const bool relu1 = a.relu1 || DST == DFX_U8;

  const int npx = th * tw;
  const int tiles_per_row = (tw + 31) >> 5;
  const int ntiles = g.linear ? (npx + 31) >> 5 : th * tiles_per_row;

  for (int t = wave; t < ntiles; t += MFMA_THREADS / 64) {
    // pixel of this lane (conv0 column) and the tile's output base
    int ty, tx, nvalid;
    size_t obase;  // dst pixel index of px_local == 0
    if (g.linear) {
      nvalid = min(32, npx - 32 * t);
      const int pc = 32 * t + min(l31, nvalid - 1);
      ty = pc / tw;
      tx = pc - ty * tw;
      obase = ((size_t)n * a.oh + y0) * a.ow + 32 * t;
    } else {
      const int tr = t / tiles_per_row, tc = t - tr * tiles_per_row;
      nvalid = min(32, tw - 32 * tc);
      ty = tr;
      tx = 32 * tc + min(l31, nvalid - 1);
      obase = ((size_t)n * a.oh + y0 + tr) * a.ow + x0 + 32 * tc;
    }
    const int Pb = ty * LW + tx;
    // Lane-constant LDS offsets are made opaque once per tile: otherwise LICM
    // hoists every weight / constant fragment read out of the tile loop and
    // keeps >200 VGPRs live across it (spills).
    int lane16 = lane * 16, h4 = 4 * h, lch = G * l31;
    asm volatile("" : "+v"(lane16), "+v"(h4), "+v"(lch));

    // ---- conv0: 9 taps x ICB k-steps x OCB row blocks ----
    v16i acc0[OCB];
#pragma unroll
    for (int r = 0; r < OCB; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4i c = *reinterpret_cast<const v4i *>(comp0 + 32 * r + 8 * q + h4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc0[r][4 * q + i] = c[i];
      }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int P = Pb + kh * LW + kw;
        const int sw = chunk_swizzle<CP>(P);
        const unsigned char *base = ins + P * IC;
#pragma unroll
        for (int c = 0; c < ICB; ++c) {
          const v4i b = *reinterpret_cast<const v4i *>(base + 16 * ((2 * c + h) ^ sw));
#pragma unroll
          for (int r = 0; r < OCB; ++r) {
            const v4i w = *reinterpret_cast<const v4i *>(
                w0s + ((r * 9 + kh * 3 + kw) * ICB + c) * 1024 + lane16);
            acc0[r] = mfma_i8(w, b, acc0[r]);
          }
        }
      }

    // ---- requant 0 in registers -> A fragments of the 1x1 ----
    v4i mid[OCB];
#pragma unroll
    for (int r = 0; r < OCB; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * r + 8 * q + h4;
        const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
        const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
        unsigned pk = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = requant(acc0[r][4 * q + i], bs[i], sc[i], true);
          pk |= sat_u8_bits(cvt_x86_rt(f, a.rm0)) << (8 * i);
        }
        mid[r][q] = (int)(pk ^ 0x80808080u);
      }

    // ---- conv1 + requant 1 + store, G column blocks at a time ----
    for (int cg = 0; cg < NCG; ++cg) {
      const int chb = 32 * G * cg + lch;  // this lane's first channel in the group
      v16i acc1[G];
      float bs[G], sc[G];
#pragma unroll
      for (int cc = 0; cc < G; ++cc) {
        const int c1 = comp1[chb + cc];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[cc][e] = c1;
        bs[cc] = bias1[chb + cc];
        sc[cc] = scale1[chb + cc];
      }
#pragma unroll
      for (int r = 0; r < OCB; ++r)
#pragma unroll
        for (int cc = 0; cc < G; ++cc) {
          const v4i w = *reinterpret_cast<const v4i *>(
              w1s + ((cg * G + cc) * OCB + r) * 1024 + lane16);
          acc1[cc] = mfma_i8(mid[r], w, acc1[cc]);
        }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pl = 8 * (e >> 2) + 4 * h + (e & 3);  // pixel (MFMA row) of register e
        if (pl < nvalid) {
          float f[G];
#pragma unroll
          for (int cc = 0; cc < G; ++cc) f[cc] = requant(acc1[cc][e], bs[cc], sc[cc], relu1);
          store_group<DST, G>(a.dst, (obase + pl) * OC1 + chb, f, a.rm1);
        }
      }
    }
  }
}

}  // namespace dfx
