// conv_stream.cuh -- general-shape u8 x s8 conv (+ReLU) [+ conv1x1 (+ReLU)] as int8-MFMA
// implicit GEMMs with STREAMED weights (gfx950 / CDNA4).
//
// conv_mfma.cuh keeps every weight resident in LDS and therefore stops at 64 channels,
// 3x3, stride 1.  This kernel covers the rest of what the reference's blocking
// admits (/root/reference/src/jit_conv_kernel.cc:512-673: ic, oc, oc1x1 multiples of 16,
// any kernel size / stride / padding; multi-chunk accumulation :193-216, :27-48):
//
//  * A workgroup of 4 waves owns a UNIT of up to 128 output pixels: a th x tw patch of
//    one image or, for small images, several whole images.  Wave w owns the 32 pixel
//    slots 32w .. 32w+31 for both contractions.
//  * K is walked in STEPS of two 32-deep MFMA k-blocks.  The packed weight fragments
//    of a step (2 x OCC or 2 x G KB, laid out by the host in exactly the order the
//    kernel walks them) travel global -> registers -> LDS, double buffered: the loads
//    of step t+1 are issued before the MFMAs of step t and written to the other
//    buffer after them; one workgroup barrier per step.  The four waves share every
//    weight fragment, so L2 sees each weight byte once per 128 pixels.
//  * The input halo tile sits in LDS one 64-channel chunk at a time ([position][64 B],
//    stored as u8 - 128, 16-byte chunks XOR-swizzled like conv_mfma.cuh).
//  * conv0 accumulates OCC 32-channel blocks at a time (D0[oc][px], weights = A
//    operand); after the last input chunk the block is requantised (ReLU, scale,
//    round, saturate to u8) and written, in the 1x1 stage's k order, to the wave's
//    own rows of the LDS intermediate mid[slot][oc] -- the reference's xmm-resident
//    intermediate (jit_conv_kernel.cc:275-277), here at most 128 x 528 bytes.
//  * conv1 runs D1[px][oc1] over mid with G column blocks at a time and the same
//    channel permutation / store path as conv_mfma.cuh (lane = G consecutive
//    channels).  The unfused op uses that orientation for the first conv directly.
//  * u8 -> s8 offset: the compensation 128 * sum(w) is added to the raw accumulator
//    as an INTEGER (K can reach 9 * 512 here, beyond f32's exact range), then the
//    reference's float(acc) (+bias) * scale chain runs unchanged.
#pragma once

#include "conv_mfma.cuh"

namespace dfx {

constexpr int ST_THREADS = 256;
constexpr int ST_M = 128;  // pixel slots per unit
constexpr int ST_TQ = 4;   // tile granules (16 B) a thread stages with precomputed addresses

struct StreamGeom {
  int ni, thv, twv;     // unit = ni whole images (ni > 1 only if thv == oh && twv == ow) x thv x twv px
  int uy, ux;           // units per image (group) along y / x
  int total_units;
  int lh, lw, npos;     // halo tile rows / cols per image; LDS positions = ni * lh * lw
  int n_icc;            // 64-channel input chunks
  int icb;              // 32-channel input blocks (ic rounded up)
  int n_occ;            // conv0 output chunks of OCC blocks
  int ocb;              // = n_occ * OCC: padded conv0 output blocks
  int n_g1, ks2;        // conv1: groups of G column blocks; k-steps (pairs of oc blocks)
  int s0_steps;         // conv0 steps per unit
  int mid_stride;       // bytes per slot of the intermediate (32 * ocb + 16)
  int off_tile, off_pxoff, off_mid;  // LDS byte offsets (weight buffers at 0)
};

// The LDS load feeding an MFMA operand must not be overtaken / re-targeted while the
// MFMA is in flight (see conv_mfma.cuh): every step loads all its fragments into
// distinct registers, fences, issues the MFMAs, fences.
#define DFX_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int OCC, int G, int DST, bool FUSED>
__global__ __launch_bounds__(ST_THREADS, 2) void conv_stream_kernel(ConvArgs a, StreamGeom g) {
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  constexpr int WB = FUSED ? (OCC > G ? OCC : G) : OCC;  // fragments per half step a buffer holds
  constexpr int WBUF = 2 * WB * 1024;                    // bytes per weight buffer
  constexpr int GA = 2 * OCC * 64, GB = 2 * G * 64;      // 16-byte granules per conv0 / conv1 step
  constexpr int NLD = (2 * WB * 64 + ST_THREADS - 1) / ST_THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *tile = smem + g.off_tile;
  int *pxoff = reinterpret_cast<int *>(smem + g.off_pxoff);
  unsigned char *mid = smem + g.off_mid;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int OCP = 32 * g.ocb, OC1P = FUSED ? 32 * G * g.n_g1 : 0;
  const int *comp0 = reinterpret_cast<const int *>(a.consts);
  const float *bias0 = a.consts + OCP, *scale0 = a.consts + 2 * OCP;
  const int *comp1 = reinterpret_cast<const int *>(a.consts + 3 * OCP);
  const float *bias1 = a.consts + 3 * OCP + OC1P, *scale1 = a.consts + 3 * OCP + 2 * OC1P;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int S0 = g.s0_steps, S1 = FUSED ? g.n_g1 * g.ks2 : 0, S = S0 + S1;
  const v4i *wsrc = reinterpret_cast<const v4i *>(a.wei);
  const int lane16 = lane * 16;
  const unsigned row_bytes = (unsigned)(FUSED ? a.oc1 : a.oc) * ESZ;

  // ---- weight stream: step t of a unit sits at granule woff(t) of the packed buffer ----
  v4i wreg[NLD];
#define DFX_W_ISSUE(T)                                                                  \
  do {                                                                                  \
    const int t_ = (T);                                                                 \
    const int off_ = t_ < S0 ? t_ * GA : S0 * GA + (t_ - S0) * GB;                      \
    const int cnt_ = t_ < S0 ? GA : GB;                                                 \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                   \
      const int q_ = tid + ST_THREADS * i;                                              \
      if (q_ < cnt_) wreg[i] = wsrc[off_ + q_];                                         \
    }                                                                                   \
  } while (0)
#define DFX_W_COMMIT(T, BUF)                                                            \
  do {                                                                                  \
    const int cnt_ = (T) < S0 ? GA : GB;                                                \
    v4i *d_ = reinterpret_cast<v4i *>(smem + (BUF) * WBUF);                             \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                   \
      const int q_ = tid + ST_THREADS * i;                                              \
      if (q_ < cnt_) d_[q_] = wreg[i];                                                  \
    }                                                                                   \
  } while (0)

  // ---- tile staging: granule q = tid + 256 i covers LDS position q >> 2, 16-byte
  //      chunk q & 3; its image / row / column inside the halo tile never change ----
  const int lhw = g.lh * g.lw;
  const int tile_q = g.npos * 4;
  int tq_rel[ST_TQ], tq_lds[ST_TQ], tq_pos[ST_TQ];  // src offset rel. to the tile origin, LDS offset, img<<20|ly<<10|lx
#pragma unroll
  for (int i = 0; i < ST_TQ; ++i) {
    const int q = tid + ST_THREADS * i;
    const int pos = min(q >> 2, g.npos - 1), j = q & 3;
    const int img = pos / lhw, r = pos - img * lhw;
    const int ly = r / g.lw, lx = r - ly * g.lw;
    tq_rel[i] = ((img * a.ih + ly) * a.iw + lx) * a.ic + 16 * j;
    tq_lds[i] = pos * 64 + 16 * (j ^ chunk_swizzle<4>(pos));
    tq_pos[i] = (img << 20) | (ly << 10) | lx;
  }
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};

  int buf = 0;
  DFX_W_ISSUE(0);
  DFX_W_COMMIT(0, 0);

  const int upg = g.uy * g.ux;
  for (int unit = blockIdx.x; unit < g.total_units; unit += gridDim.x) {
    const int grp = unit / upg, u = unit - grp * upg;
    const int uyi = u / g.ux, uxi = u - uyi * g.ux;
    const int n0 = grp * g.ni, y0 = uyi * g.thv, x0 = uxi * g.twv;
    const int nimg = min(g.ni, a.bs - n0), thc = min(g.thv, a.oh - y0), twc = min(g.twv, a.ow - x0);
    const int npx = nimg * thc * twc;
    // this lane's pixel slot (conv0 column / conv1 row)
    const int slot = 32 * wave + l31;
    int Pb;
    {
      const int pc = min(slot, npx - 1);
      const int img = pc / (thc * twc), r = pc - img * (thc * twc);
      const int ty = r / twc, tx = r - ty * twc;
      Pb = img * lhw + ty * a.sh * g.lw + tx * a.sw;
      if (h == 0) pxoff[slot] = slot < npx ? ((n0 + img) * a.oh + y0 + ty) * a.ow + x0 + tx : -1;
    }
    const int iy0 = y0 * a.sh - a.pt, ix0 = x0 * a.sw - a.pl;
    // origin of the halo tile in src (may point before the image: only used with valid offsets)
    const long long org = (((long long)n0 * a.ih + iy0) * a.iw + ix0) * a.ic;
    unsigned char *my_mid = mid + slot * g.mid_stride + h * 16;

    int t = 0;
    for (int occ = 0; occ < g.n_occ; ++occ) {
      v16i acc[OCC];
#pragma unroll
      for (int r = 0; r < OCC; ++r) acc[r] = zero16;
      for (int icc = 0; icc < g.n_icc; ++icc) {
        // ---- stage input chunk icc of the halo tile (all waves are past the last
        //      step that read the previous contents: every step ends in a barrier).
        //      A single-chunk input stays in LDS for all output chunks of the unit. ----
        if (g.n_icc > 1 || occ == 0) {
          const int cb0 = 64 * icc;
          v4i tv[ST_TQ];
#pragma unroll
          for (int i = 0; i < ST_TQ; ++i) {
            const int q = tid + ST_THREADS * i;
            const int img = tq_pos[i] >> 20, ly = (tq_pos[i] >> 10) & 1023, lx = tq_pos[i] & 1023;
            const int iy = iy0 + ly, ix = ix0 + lx;
            const bool ok = q < tile_q && img < nimg && iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw &&
                            cb0 + 16 * (q & 3) < a.ic;
            tv[i] = v4i{0, 0, 0, 0};
            if (ok) tv[i] = *reinterpret_cast<const v4i *>(a.src + (org + tq_rel[i] + cb0));
          }
#pragma unroll
          for (int i = 0; i < ST_TQ; ++i)
            if (tid + ST_THREADS * i < tile_q) *reinterpret_cast<v4i *>(tile + tq_lds[i]) = tv[i] ^ x80;
          for (int q = tid + ST_THREADS * ST_TQ; q < tile_q; q += ST_THREADS) {  // oversized tiles
            const int pos = q >> 2, j = q & 3;
            const int img = pos / lhw, r = pos - img * lhw;
            const int ly = r / g.lw, lx = r - ly * g.lw;
            const int iy = iy0 + ly, ix = ix0 + lx;
            const bool ok = img < nimg && iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw && cb0 + 16 * j < a.ic;
            v4i v = v4i{0, 0, 0, 0};
            if (ok)
              v = *reinterpret_cast<const v4i *>(
                  a.src + (org + (long long)((img * a.ih + ly) * a.iw + lx) * a.ic + 16 * j + cb0));
            *reinterpret_cast<v4i *>(tile + pos * 64 + 16 * (j ^ chunk_swizzle<4>(pos))) = v ^ x80;
          }
          __syncthreads();
        }
        const int kbn = min(2, g.icb - 2 * icc);  // 32-channel blocks in this chunk
        const int ntap = a.kh * a.kw;
        const int ns = ntap * kbn, ns2 = (ns + 1) >> 1;
        int tkh = 0, tkw = 0;  // tap of k-block s (kept incrementally)
        for (int s2 = 0; s2 < ns2; ++s2) {
          int tn = t + 1;
          if (tn == S) tn = 0;
          DFX_W_ISSUE(tn);
          const unsigned char *wb = smem + buf * WBUF;
          v4i fb[2], fw[2][OCC];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int s = 2 * s2 + j;
            const int icbl = kbn == 2 ? j : 0;
            const int P = Pb + tkh * g.lw + tkw;
            fb[j] = *reinterpret_cast<const v4i *>(tile + P * 64 + 16 * ((2 * icbl + h) ^ chunk_swizzle<4>(P)));
#pragma unroll
            for (int r = 0; r < OCC; ++r)
              fw[j][r] = *reinterpret_cast<const v4i *>(wb + (j * OCC + r) * 1024 + lane16);
            // advance the tap after the chunk's last k-block of it; a padding k-block
            // (s >= ns, zero weights) re-reads the last tap
            if ((kbn == 1 || j == 1) && s + 1 < ns) {
              if (++tkw == a.kw) { tkw = 0; ++tkh; }
            }
          }
          DFX_FENCE();
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < OCC; ++r)
              acc[r] = FUSED ? mfma_i8(fw[j][r], fb[j], acc[r])   // D0[oc][px]
                             : mfma_i8(fb[j], fw[j][r], acc[r]);  // D0[px][oc]
          DFX_FENCE();
          DFX_W_COMMIT(tn, buf ^ 1);
          __syncthreads();
          buf ^= 1;
          ++t;
        }
      }
      if constexpr (FUSED) {
        // ---- requant 0 -> u8 -> this wave's rows of mid, in the 1x1 stage's k order:
        //      byte 16h + 4q + i of block r  =  channel 32r + 8q + 4h + i ----
#pragma unroll
        for (int r = 0; r < OCC; ++r) {
          v4i pkv;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ch = (occ * OCC + r) * 32 + 8 * q + 4 * h;
            const v4i cp = *reinterpret_cast<const v4i *>(comp0 + ch);
            const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
            const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float f = requant(acc[r][4 * q + i] + cp[i], bs[i], sc[i], true);
              pk |= sat_u8_bits(cvt_x86_rt(f, a.rm0)) << (8 * i);
            }
            pkv[q] = (int)(pk ^ 0x80808080u);
          }
          *reinterpret_cast<v4i *>(my_mid + (occ * OCC + r) * 32) = pkv;
        }
      } else {
        // ---- unfused: typed store; lane owns channels 32*OCC*occ + OCC*l31 + {0..OCC-1} ----
        const int chb = 32 * OCC * occ + OCC * l31;
        if (chb < a.oc) {
          int cp[OCC];
          float bs[OCC], sc[OCC], zf[OCC];
#pragma unroll
          for (int cc = 0; cc < OCC; ++cc) {
            cp[cc] = comp0[chb + cc];
            bs[cc] = bias0[chb + cc];
            sc[cc] = scale0[chb + cc];
            zf[cc] = 0.0f;
          }
          const bool relu = a.relu0 || DST == DFX_U8;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int off = pxoff[32 * wave + 8 * (e >> 2) + (e & 3) + 4 * h];
            if (off >= 0) {
              int v[OCC];
#pragma unroll
              for (int cc = 0; cc < OCC; ++cc) v[cc] = acc[cc][e] + cp[cc];
              store_group<DST, OCC, false>(reinterpret_cast<unsigned char *>(a.dst) + (size_t)off * row_bytes +
                                               (unsigned)chb * ESZ,
                                           v, zf, bs, sc, relu, a.rm0);
            }
          }
        }
      }
    }

    if constexpr (FUSED) {
      // ---- conv1 over mid, G column blocks at a time ----
      const bool relu = a.relu1 || DST == DFX_U8;
      for (int g1 = 0; g1 < g.n_g1; ++g1) {
        v16i acc1[G];
#pragma unroll
        for (int cc = 0; cc < G; ++cc) acc1[cc] = zero16;
        for (int s2 = 0; s2 < g.ks2; ++s2) {
          int tn = t + 1;
          if (tn == S) tn = 0;
          DFX_W_ISSUE(tn);
          const unsigned char *wb = smem + buf * WBUF;
          v4i fa[2], fw[2][G];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int blk = min(2 * s2 + j, g.ocb - 1);  // a padding k-block has zero weights
            fa[j] = *reinterpret_cast<const v4i *>(my_mid + blk * 32);
#pragma unroll
            for (int cc = 0; cc < G; ++cc)
              fw[j][cc] = *reinterpret_cast<const v4i *>(wb + (j * G + cc) * 1024 + lane16);
          }
          DFX_FENCE();
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int cc = 0; cc < G; ++cc) acc1[cc] = mfma_i8(fa[j], fw[j][cc], acc1[cc]);
          DFX_FENCE();
          DFX_W_COMMIT(tn, buf ^ 1);
          __syncthreads();
          buf ^= 1;
          ++t;
        }
        const int chb = 32 * G * g1 + G * l31;
        if (chb < a.oc1) {
          int cp[G];
          float bs[G], sc[G], zf[G];
#pragma unroll
          for (int cc = 0; cc < G; ++cc) {
            cp[cc] = comp1[chb + cc];
            bs[cc] = bias1[chb + cc];
            sc[cc] = scale1[chb + cc];
            zf[cc] = 0.0f;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int off = pxoff[32 * wave + 8 * (e >> 2) + (e & 3) + 4 * h];
            if (off >= 0) {
              int v[G];
#pragma unroll
              for (int cc = 0; cc < G; ++cc) v[cc] = acc1[cc][e] + cp[cc];
              store_group<DST, G, false>(reinterpret_cast<unsigned char *>(a.dst) + (size_t)off * row_bytes +
                                             (unsigned)chb * ESZ,
                                         v, zf, bs, sc, relu, a.rm1);
            }
          }
        }
      }
    }
  }
#undef DFX_W_ISSUE
#undef DFX_W_COMMIT
}

}  // namespace dfx
