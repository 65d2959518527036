// conv_direct.cuh -- fused u8 x s8 conv (+ReLU) + conv1x1 (+ReLU) for general shapes as two
// int8-MFMA implicit GEMMs whose WEIGHT fragments go straight from global memory (L2) into
// the MFMA operand registers of the one wave that needs them (gfx950 / CDNA4).
//
// Same contract as conv_stream.cuh (/root/reference/src/jit_conv_kernel.cc:143-393 with the
// multi-chunk accumulation of :193-216, any kernel size / stride / padding, channels
// multiples of 16), different decomposition.  conv_stream.cuh gives every wave 32 pixels and
// ALL output channels, so the four waves share each weight fragment through LDS: a staging
// copy and a workgroup barrier per 8 MFMAs.  Here a workgroup still owns a unit of up to 128
// output pixels, but
//  * in conv0 wave w owns output-channel blocks w, w+WO, ... for PXW = WO of the unit's four
//    32-pixel blocks (WO x WP waves, WO * WP = 4).  Its weight fragments are private: they are
//    fetched with one coalesced 1 KB global load per 32-deep k-block into a register ring
//    DK_RD k-blocks deep (the L2 latency is ~1k cycles, a k-block is PXW MFMAs), and feed the
//    MFMA directly as the A operand.  No weight staging, no barrier inside the K loop.
//  * the input halo tile is staged ONCE per unit with all its 64-channel planes
//    ([plane][position][80 B]: the odd multiple of 16 spreads consecutive positions over the
//    banks, so a fragment address is base + a wave-uniform offset), read-only afterwards;
//  * the u8 intermediate goes to LDS (mid[slot][oc], the reference keeps it in xmm registers,
//    jit_conv_kernel.cc:275-277); one barrier; then conv1 with the same idea: wave w owns
//    1x1 output groups w, w+WO1, ... (G column blocks each, the channel permutation and
//    store path of conv_mfma.cuh), A fragments from mid, B fragments from global.
// Three workgroup barriers per unit in all.  Requantisation, fast/exact paths, the LDS
// transpose for 1-byte outputs and the unit geometry are those of conv_stream.cuh.
#pragma once

#include "conv_mfma.cuh"

namespace dfx {

constexpr int DK_THREADS = 256;
constexpr int DK_M = 128;    // pixel slots per unit
constexpr int DK_POS = 80;   // LDS bytes per halo-tile position and plane (64 + 16 pad)
constexpr int DK_TQ = 4;     // tile granules a thread prefetches into registers
constexpr int DK_RD = 9;     // conv0 weight ring: k-blocks in flight per wave (multiple of 3)
constexpr int DK_STAGE = 32 * 144;  // per wave: 1-byte store staging (aliases the dead tile)

struct DirectGeom {
  int ni, thv, twv;     // unit = ni whole images (ni > 1 only if thv == oh && twv == ow) x thv x twv px
  int uy, ux, total_units;
  int lh, lw, npos;     // halo tile rows / cols per image; positions = ni * lh * lw
  int icb;              // 32-channel input blocks
  int n_planes;         // 64-channel planes of the tile = (icb + 1) / 2
  int plane_bytes;      // npos * DK_POS
  int ocb;              // conv0 output blocks, padded to a multiple of WO
  int n_g1;             // conv1 groups of G column blocks
  int mid_stride;       // 32 * ocb + 16
  int off_pxoff, off_mid, off_cst;  // LDS byte offsets (tile and the aliased staging at 0)
  int fast;             // 1: fast requant path valid (host proof)
#ifdef DFX_STAMPS
  unsigned long long *prof;  // diagnostic build only: [workgroup][wave][16] cycle sums
#endif
#ifdef DK_DEBUG
  // bounds-checking diagnostic build: every global access is checked against these sizes; the
  // first violation per tag is recorded in dbg[2 tag] (offset) / dbg[2 tag + 1] (size) and the
  // access is redirected to offset 0
  long long src_bytes, dst_bytes, wei_bytes, wei1_bytes, cst_bytes;
  long long *dbg;
#endif
};

#define DKF() __builtin_amdgcn_sched_barrier(0)
#ifdef DK_DEBUG
#define DK_CHK(TAG, OFF, LEN, SIZE)                                                     \
  ([&]() -> long long {                                                                 \
    const long long o__ = (long long)(OFF);                                             \
    if (o__ < 0 || o__ + (LEN) > (SIZE)) {                                              \
      g.dbg[2 * (TAG)] = o__;                                                           \
      g.dbg[2 * (TAG) + 1] = (SIZE);                                                    \
      return 0ll;                                                                       \
    }                                                                                   \
    return o__;                                                                         \
  }())
#else
#define DK_CHK(TAG, OFF, LEN, SIZE) (OFF)
#endif

template <int WO, int G, int WO1, int DST>
__global__ __launch_bounds__(DK_THREADS, 2) void conv_direct_kernel(ConvArgs a, DirectGeom g) {
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  constexpr int WP = 4 / WO, PXW = WO;        // conv0: WO x WP waves, PXW pixel blocks per wave
  constexpr int WP1 = 4 / WO1, PXW1 = WO1;    // conv1 likewise
  constexpr int PX1 = PXW1 > 2 ? 2 : PXW1;    // pixel blocks per conv1 pass (accumulator budget)
  constexpr int NP1 = PXW1 / PX1;             // conv1 passes per group
  constexpr int RD1 = 4;                      // conv1 weight ring depth in k-blocks (G fragments each)
  constexpr int NF = PXW < 2 ? PXW : 2;       // MFMAs issued before the k-block's LDS prefetch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const tile0 = smem;
  unsigned *pxoff = reinterpret_cast<unsigned *>(smem + g.off_pxoff);
  unsigned char *mid = smem + g.off_mid;
  float *cst0 = reinterpret_cast<float *>(smem + g.off_cst);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wo = wave / WP, wp = wave % WP, wo1 = wave / WP1, wp1 = wave % WP1;
  const int OCP = 32 * g.ocb, OC1P = 32 * G * g.n_g1;
  const int *comp0 = reinterpret_cast<const int *>(cst0);
  const float *bias0 = cst0 + OCP, *scale0 = cst0 + 2 * OCP;
  const int *comp1 = reinterpret_cast<const int *>(a.consts + 3 * OCP);
  const float *bias1 = a.consts + 3 * OCP + OC1P, *scale1 = a.consts + 3 * OCP + 2 * OC1P;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int ntap = a.kh * a.kw, nkb0 = g.icb * ntap;
  // weights: a.wei = W0d[ocb][nkb0][64 lanes][16 B], a.wei1 = W1d[n_g1][ocb][G][64 lanes][16 B]
  const unsigned row_bytes = (unsigned)a.oc1 * ESZ;
  const bool fast = g.fast != 0;
  const bool relu1 = a.relu1 || DST == DFX_U8;
  using TT = std::true_type;
  using FF = std::false_type;

  // ---- tile staging: granule q = tid + 256 i -> plane q / (4 npos), position, 16-byte chunk q & 3 ----
  const int lhw = g.lh * g.lw;
  const int row_skip = (g.lw - a.kw) * DK_POS;  // bytes from the last tap of a kernel row to the next row's first
  const int tile_q1 = g.npos * 4, tile_q = tile_q1 * g.n_planes;
  int tq_pos[DK_TQ];  // plane << 28 | img << 20 | ly << 10 | lx
#pragma unroll
  for (int i = 0; i < DK_TQ; ++i) {
    const int q = min(tid + DK_THREADS * i, tile_q - 1);
    const int pl = q / tile_q1, pos = (q - pl * tile_q1) >> 2;
    const int img = pos / lhw, r = pos - img * lhw;
    const int ly = r / g.lw, lx = r - ly * g.lw;
    tq_pos[i] = (pl << 28) | (img << 20) | (ly << 10) | lx;
  }
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
  v4i tv[DK_TQ];
  int tv_ok = 0;
  // (branch-free loads: hipcc waits vmcnt(0) inside a branch around a load)
#define DK_T_ISSUE(N0, IY0, IX0, NIMG)                                                  \
  do {                                                                                  \
    _Pragma("unroll") for (int i = 0; i < DK_TQ; ++i) {                                 \
      const int q_ = tid + DK_THREADS * i;                                              \
      const int pl_ = (tq_pos[i] >> 28) & 15, img_ = (tq_pos[i] >> 20) & 255;           \
      const int ly_ = (tq_pos[i] >> 10) & 1023, lx_ = tq_pos[i] & 1023;                 \
      const int iy_ = (IY0) + ly_, ix_ = (IX0) + lx_, cb_ = 64 * pl_ + 16 * (q_ & 3);   \
      const bool ok_ = q_ < tile_q && img_ < (NIMG) && iy_ >= 0 && iy_ < a.ih && ix_ >= 0 && \
                       ix_ < a.iw && cb_ < a.ic;                                        \
      /* always an in-range address (clamped coordinates); padding is zeroed at commit */ \
      const int n_ = min((N0) + img_, a.bs - 1), y_ = min(max(iy_, 0), a.ih - 1);       \
      const int x_ = min(max(ix_, 0), a.iw - 1), c_ = min(cb_, a.ic - 16);              \
      const long long o_ = (((long long)n_ * a.ih + y_) * a.iw + x_) * a.ic + c_;       \
      tv[i] = *reinterpret_cast<const v4i *>(a.src + DK_CHK(1, o_, 16, g.src_bytes));   \
      tv_ok = ok_ ? (tv_ok | (1 << i)) : (tv_ok & ~(1 << i));                           \
    }                                                                                   \
  } while (0)
#define DK_T_COMMIT()                                                                   \
  do {                                                                                  \
    _Pragma("unroll") for (int i = 0; i < DK_TQ; ++i) {                                 \
      const int q_ = tid + DK_THREADS * i;                                              \
      const int pl_ = (tq_pos[i] >> 28) & 15, img_ = (tq_pos[i] >> 20) & 255;           \
      const int ly_ = (tq_pos[i] >> 10) & 1023, lx_ = tq_pos[i] & 1023;                 \
      const int lo_ = q_ < tile_q ? pl_ * g.plane_bytes + (img_ * lhw + ly_ * g.lw + lx_) * DK_POS + 16 * (q_ & 3) \
                                  : g.n_planes * g.plane_bytes; /* dump slot */          \
      *reinterpret_cast<v4i *>(tile0 + lo_) = ((tv_ok >> i) & 1) ? tv[i] ^ x80 : x80;   \
    }                                                                                   \
  } while (0)

  const int upg = g.uy * g.ux;
  struct UnitGeo { int n0, y0, x0, nimg, iy0, ix0; };
  auto unit_geo = [&](int unit) {
    UnitGeo r;
    const int grp = unit / upg, u = unit - grp * upg;
    const int uyi = u / g.ux, uxi = u - uyi * g.ux;
    r.n0 = grp * g.ni; r.y0 = uyi * g.thv; r.x0 = uxi * g.twv;
    r.nimg = min(g.ni, a.bs - r.n0);
    r.iy0 = r.y0 * a.sh - a.pt; r.ix0 = r.x0 * a.sw - a.pl;
    return r;
  };

  for (int q = tid; q < 3 * OCP; q += DK_THREADS)  // visible after the first barrier
    cst0[q] = a.consts[DK_CHK(3, (long long)q * 4, 4, g.cst_bytes) / 4];
#ifdef DFX_STAMPS
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

  for (int unit = blockIdx.x; unit < g.total_units; unit += gridDim.x) {
    const UnitGeo ug = unit_geo(unit);
    const int thc = min(g.thv, a.oh - ug.y0), twc = min(g.twv, a.ow - ug.x0);
    const int npx = ug.nimg * thc * twc;
    DFX_STAMP(t0);
    // every wave is out of the previous unit (its store staging aliases the tile; pxoff, mid)
    __syncthreads();
    // ---- stage the whole halo tile (all planes) ----
    DK_T_ISSUE(ug.n0, ug.iy0, ug.ix0, ug.nimg);
    {  // slot table: wave w fills pixel block w
      const int slot = 32 * wave + l31;
      const int pc = min(slot, npx - 1);
      const int img = pc / (thc * twc), r = pc - img * (thc * twc);
      const int ty = r / twc, tx = r - ty * twc;
      if (h == 0)
        pxoff[slot] = slot < npx ? (unsigned)(((ug.n0 + img) * a.oh + ug.y0 + ty) * a.ow + ug.x0 + tx) * row_bytes
                                 : 0xffffffffu;
    }
    DK_T_COMMIT();
    for (int q = tid + DK_THREADS * DK_TQ; q < tile_q; q += DK_THREADS) {  // the part beyond the register prefetch
      const int pl = q / tile_q1, ql = q - pl * tile_q1;
      const int pos = ql >> 2, j = ql & 3;
      const int img = pos / lhw, r = pos - img * lhw;
      const int ly = r / g.lw, lx = r - ly * g.lw;
      const int iy = ug.iy0 + ly, ix = ug.ix0 + lx;
      const bool ok = img < ug.nimg && iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw && 64 * pl + 16 * j < a.ic;
      const int n_ = min(ug.n0 + img, a.bs - 1), y_ = min(max(iy, 0), a.ih - 1), x_ = min(max(ix, 0), a.iw - 1);
      const long long o = (((long long)n_ * a.ih + y_) * a.iw + x_) * a.ic + min(64 * pl + 16 * j, a.ic - 16);
      const v4i v = *reinterpret_cast<const v4i *>(a.src + DK_CHK(2, o, 16, g.src_bytes));
      *reinterpret_cast<v4i *>(tile0 + pl * g.plane_bytes + pos * DK_POS + 16 * j) = ok ? v ^ x80 : x80;
    }
    __syncthreads();

    DFX_STAMP(t1);
    DFX_ACC(0, t1 - t0);  // barrier + tile staging
    // ---- conv0: this wave's output blocks x its PXW pixel blocks ----
    int fbyte[PXW];  // tile byte offset of the slot's input position (tap 0, plane 0) + this lane's k half
    unsigned char *mid_w[PXW];
#pragma unroll
    for (int p = 0; p < PXW; ++p) {
      const int slot = 32 * (wp * PXW + p) + l31;
      const int pc = min(slot, npx - 1);
      const int img = pc / (thc * twc), r = pc - img * (thc * twc);
      const int ty = r / twc, tx = r - ty * twc;
      fbyte[p] = (img * lhw + ty * a.sh * g.lw + tx * a.sw) * DK_POS + 16 * h;
      mid_w[p] = mid + slot * g.mid_stride + h * 16;
    }
    for (int ob = wo; ob < g.ocb; ob += WO) {
      DFX_STAMP(t1b);
      v16i acc[PXW];
#pragma unroll
      for (int p = 0; p < PXW; ++p) acc[p] = zero16;
      v4i wr[DK_RD];
#pragma unroll
      for (int i = 0; i < DK_RD; ++i)
        wr[i] = *reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(a.wei) + DK_CHK(4, ((long long)ob * nkb0 + min(i, nkb0 - 1)) * 1024 + lane * 16, 16, g.wei_bytes));
      // position of the NEXT k-block whose pixel fragments get loaded (runs two ahead)
      int l_tap = 0, l_tkw = 0, l_toff = 0, l_icb = 0;
      v4i fb[3][PXW];
#define DK_LOAD_FB(SET)                                                                 \
  do {                                                                                  \
    const int koff_ = __builtin_amdgcn_readfirstlane((l_icb >> 1) * g.plane_bytes + (l_icb & 1) * 32 + l_toff); \
    _Pragma("unroll") for (int p = 0; p < PXW; ++p)                                     \
      fb[SET][p] = *reinterpret_cast<const v4i *>(tile0 + fbyte[p] + koff_);            \
    l_toff += DK_POS;                                                                   \
    if (++l_tkw == a.kw) { l_tkw = 0; l_toff += row_skip; }                             \
    if (++l_tap == ntap) { l_tap = 0; l_tkw = 0; l_toff = 0; ++l_icb; }                 \
  } while (0)
      DK_LOAD_FB(0);
      if (nkb0 > 1) DK_LOAD_FB(1);
      DKF();
      for (int kb0 = 0; kb0 < nkb0; kb0 += DK_RD) {
#pragma unroll
        for (int i = 0; i < DK_RD; ++i) {
          const int kb = kb0 + i;
          if (kb < nkb0) {
#pragma unroll
            for (int p = 0; p < NF; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], acc[p]);  // D0[oc][px]
            DKF();
            if (kb + 2 < nkb0) DK_LOAD_FB((i + 2) % 3);
            DKF();
#pragma unroll
            for (int p = NF; p < PXW; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], acc[p]);
            DKF();
            // refill the ring slot (returns ~1k cycles later)
            wr[i] = *reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(a.wei) + DK_CHK(5, ((long long)ob * nkb0 + min(kb + DK_RD, nkb0 - 1)) * 1024 + lane * 16, 16, g.wei_bytes));
          }
        }
      }
#undef DK_LOAD_FB
      DFX_STAMP(t2);
      // requant 0 -> u8 -> mid, in the 1x1 stage's k order: byte 16h + 4q + i of block ob = channel 32 ob + 8q + 4h + i
#pragma unroll
      for (int p = 0; p < PXW; ++p) {
        v4i pkv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = ob * 32 + 8 * q + 4 * h;
          const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
          const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
          unsigned pk = 0;
          if (fast) {
#pragma unroll
            for (int i = 0; i < 4; ++i)  // plain v_add_f32 / v_mul_f32: the packed forms do not overlap with MFMAs (conv_mfma.cuh)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__fmul_rn(__fadd_rn(__int2float_rn(acc[p][4 * q + i]), bs[i]), sc[i]), i, pk);
          } else {
            const v4i cp = *reinterpret_cast<const v4i *>(comp0 + ch);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float f = requant(acc[p][4 * q + i] + cp[i], bs[i], sc[i], true);
              pk |= sat_u8_bits(cvt_x86_rt(f, a.rm0)) << (8 * i);
            }
          }
          pkv[q] = (int)(pk ^ 0x80808080u);
        }
        *reinterpret_cast<v4i *>(mid_w[p] + ob * 32) = pkv;
      }
      DFX_STAMP(t3);
      DFX_ACC(2, t3 - t2);  // requant 0
      DFX_ACC(1, t2 - t1b);  // conv0 K loop
    }
    DFX_STAMP(t4);
    __syncthreads();  // mid is complete; the tile is dead (the store staging may use it)
    DFX_STAMP(t5);
    DFX_ACC(3, t5 - t4);  // barrier after conv0

    // ---- conv1: this wave's groups x its PXW1 pixel blocks, PX1 at a time ----
    v4i wr1[RD1][G];  // W1 fragments of the next k-blocks of this wave's (group, pass) sequence
#define DK_W1_PRELOAD(G1)                                                               \
  _Pragma("unroll") for (int i = 0; i < RD1; ++i)                                       \
    _Pragma("unroll") for (int cc = 0; cc < G; ++cc)                                    \
      wr1[i][cc] = *reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(a.wei1) + \
          DK_CHK(6, (((long long)(G1) * g.ocb + min(i, g.ocb - 1)) * G + cc) * 1024 + lane * 16, 16, g.wei1_bytes))
    DK_W1_PRELOAD(wo1);
    for (int g1 = wo1; g1 < g.n_g1; g1 += WO1) {
      const int chb = 32 * G * g1 + G * l31;
      int cp[G];
      float bs[G], sc[G], zf[G];
#pragma unroll
      for (int cc = 0; cc < G; ++cc) {
        cp[cc] = fast ? 0 : comp1[DK_CHK(8, (long long)(3 * OCP + chb + cc) * 4, 4, g.cst_bytes) / 4 - 3 * OCP];
        bs[cc] = bias1[DK_CHK(9, (long long)(3 * OCP + OC1P + chb + cc) * 4, 4, g.cst_bytes) / 4 - 3 * OCP - OC1P];
        sc[cc] = scale1[DK_CHK(10, (long long)(3 * OCP + 2 * OC1P + chb + cc) * 4, 4, g.cst_bytes) / 4 - 3 * OCP - 2 * OC1P];
        zf[cc] = 0.0f;
      }
#pragma unroll
      for (int pp = 0; pp < NP1; ++pp) {
        DFX_STAMP(t6);
        const int pb0 = wp1 * PXW1 + pp * PX1;  // first pixel block of this pass
        unsigned char *mid_r[PX1];
#pragma unroll
        for (int p = 0; p < PX1; ++p) mid_r[p] = mid + (32 * (pb0 + p) + l31) * g.mid_stride + h * 16;
        v16i acc1[PX1][G];
#pragma unroll
        for (int p = 0; p < PX1; ++p)
#pragma unroll
          for (int cc = 0; cc < G; ++cc) acc1[p][cc] = zero16;
        v4i fa[2][PX1];
#pragma unroll
        for (int p = 0; p < PX1; ++p) fa[0][p] = *reinterpret_cast<const v4i *>(mid_r[p]);
        DKF();
        for (int b0 = 0; b0 < g.ocb; b0 += RD1) {
#pragma unroll
          for (int i = 0; i < RD1; ++i) {
            const int blk = b0 + i;
            if (blk < g.ocb) {
              constexpr int NM = PX1 * G, NF1 = NM < 2 ? NM : 2;
#pragma unroll
              for (int m = 0; m < NF1; ++m)
                acc1[m % PX1][m / PX1] = mfma_i8(fa[i & 1][m % PX1], wr1[i][m / PX1], acc1[m % PX1][m / PX1]);
              DKF();
              if (blk + 1 < g.ocb) {
#pragma unroll
                for (int p = 0; p < PX1; ++p)
                  fa[(i + 1) & 1][p] = *reinterpret_cast<const v4i *>(mid_r[p] + (blk + 1) * 32);
              }
              DKF();
#pragma unroll
              for (int m = NF1; m < NM; ++m)
                acc1[m % PX1][m / PX1] = mfma_i8(fa[i & 1][m % PX1], wr1[i][m / PX1], acc1[m % PX1][m / PX1]);
              DKF();
#pragma unroll
              for (int cc = 0; cc < G; ++cc)  // refill within the group (a clamped repeat at its end is harmless)
                wr1[i][cc] = *reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(a.wei1) + DK_CHK(7, (((long long)g1 * g.ocb + min(blk + RD1, g.ocb - 1)) * G + cc) * 1024 + lane * 16, 16, g.wei1_bytes));
            }
          }
        }
        DFX_STAMP(t7);
        DFX_ACC(4, t7 - t6);  // conv1 K loop
        // the ring for the next pass / group travels under this store epilogue
        if (pp + 1 < NP1) { DK_W1_PRELOAD(g1); } else if (g1 + WO1 < g.n_g1) { DK_W1_PRELOAD(g1 + WO1); }
        // ---- requant 1 + store ----
        unsigned char *dst_b = reinterpret_cast<unsigned char *>(a.dst);
        const unsigned chbE = (unsigned)chb * ESZ;
        auto emit = [&](auto fast_tag) {
          if constexpr (ESZ == 1 && G == 4) {
            // 1-byte outputs: transpose 32 px x 128 B through LDS, 16-byte stores (see conv_stream.cuh)
            unsigned char *stg = tile0 + wave * DK_STAGE;
#pragma unroll
            for (int p = 0; p < PX1; ++p) {
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                int v[G];
#pragma unroll
                for (int cc = 0; cc < G; ++cc) v[cc] = acc1[p][cc][e] + cp[cc];
                const unsigned pk = pack_group<DST, G, decltype(fast_tag)::value>(v, zf, bs, sc, relu1, a.rm1);
                *reinterpret_cast<unsigned *>(stg + (8 * (e >> 2) + (e & 3) + 4 * h) * 144 + 4 * l31) = pk;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k, px = c >> 3, c16 = c & 7;
                const unsigned off = pxoff[32 * (pb0 + p) + px];
                const v4i val = *reinterpret_cast<const v4i *>(stg + px * 144 + 16 * c16);
                if (off != 0xffffffffu && 128 * g1 + 16 * c16 < a.oc1)
                  DFX_STORE16(reinterpret_cast<v4i *>(dst_b + DK_CHK(11, (long long)(off + 128 * g1 + 16 * c16), 16, g.dst_bytes)), val);
              }
            }
          } else {
            if (chb < a.oc1) {
#pragma unroll
              for (int p = 0; p < PX1; ++p)
#pragma unroll
                for (int eq = 0; eq < 4; ++eq) {
                  const v4i o4 = *reinterpret_cast<const v4i *>(pxoff + 32 * (pb0 + p) + 8 * eq + 4 * h);
#pragma unroll
                  for (int i = 0; i < 4; ++i) {
                    const unsigned off = (unsigned)o4[i];
                    if (off != 0xffffffffu) {
                      int v[G];
#pragma unroll
                      for (int cc = 0; cc < G; ++cc) v[cc] = acc1[p][cc][4 * eq + i] + cp[cc];
                      store_group<DST, G, decltype(fast_tag)::value>(dst_b + DK_CHK(12, (long long)(off + chbE), G * ESZ, g.dst_bytes), v, zf, bs, sc,
                                                                     relu1, a.rm1);
                    }
                  }
                }
            }
          }
        };
        if (fast) emit(TT{}); else emit(FF{});
        DFX_STAMP(t8);
        DFX_ACC(5, t8 - t7);  // requant 1 + stores
      }
    }
    DFX_STAMP(t9);
    DFX_ACC(6, t9 - t0);
    DFX_ACC(7, 1);
  }
#ifdef DFX_STAMPS
  if (lane == 0) {
    unsigned long long *o = g.prof + ((size_t)blockIdx.x * 4 + wave) * 16;
    for (int k = 0; k < 8; ++k) o[k] = prof_acc[k];
  }
#endif
#undef DK_W1_PRELOAD
#undef DK_T_ISSUE
#undef DK_T_COMMIT
}

#undef DKF

}  // namespace dfx
