// conv_stream.cuh -- general-shape u8 x s8 conv (+ReLU) [+ conv1x1 (+ReLU)] as int8-MFMA
// implicit GEMMs with STREAMED weights (gfx950 / CDNA4).
//
// conv_mfma.cuh keeps every weight resident in LDS and therefore stops at 64 channels,
// 3x3, stride 1.  This kernel covers the rest of what the reference's blocking
// admits (/root/reference/src/jit_conv_kernel.cc:512-673: ic, oc, oc1x1 multiples of 16,
// any kernel size / stride / padding; multi-chunk accumulation :193-216, :27-48):
//
//  * A workgroup of 4 waves owns a UNIT of up to 128 * PXB output pixels: a th x tw patch
//    of one image or, for small images, several whole images.  Wave w owns PXB blocks of
//    32 pixel slots for both contractions; with PXB = 2 every weight fragment read from
//    LDS feeds two MFMAs (the kernel is LDS-bandwidth bound at PXB = 1: 1.25 KB of
//    fragments per MFMA against 128 B/clk/CU), at the price of twice the intermediate.
//  * K is walked in STEPS of two 32-deep MFMA k-blocks.  The packed weight fragments
//    of a step (2 x OCC or 2 x G KB, laid out by the host in exactly the order the
//    kernel walks them) travel global -> registers -> LDS through THREE LDS buffers: the
//    loads of step t+3 are issued during step t and written during step t+1 into the
//    buffer step t freed, so that step t+2's weights are already visible (one workgroup
//    barrier per step) when a wave prefetches them.  The four waves share every weight
//    fragment, so L2 sees each weight byte once per 128 pixels.  The next input chunk /
//    next unit's tile is likewise fetched into registers during the last step before
//    it is needed.
//  * Inside a phase (one input chunk of conv0, or the 1x1 stage) every wave software-
//    pipelines at k-block granularity: two fragment register sets; while the MFMAs of
//    one k-block run, the fragments of the next one (possibly of the next step) are
//    read from LDS and the weight pipeline's loads / LDS writes are issued, so a wave
//    keeps the matrix pipe busy by itself.
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
//  * Few pixels, many channels (14x14, 7x7 images): the unfused kernel can hand out (unit,
//    output chunk) work items instead of whole units (occ_par), and the host runs a fused op
//    whose units would not fill the machine as two such launches through a u8 intermediate
//    in global memory (it stays in L2) -- same arithmetic, see dfx_api.hip.
//  * u8 -> s8 offset: the compensation 128 * sum(w) is added to the raw accumulator
//    as an INTEGER (K can reach 9 * 512 here, beyond f32's exact range), then the
//    reference's float(acc) (+bias) * scale chain runs unchanged.
#pragma once

#include "conv_mfma.cuh"

namespace dfx {

constexpr int ST_THREADS = 256;
constexpr int ST_M = 128;  // pixel slots per unit and PXB
constexpr int ST_TQ = 4;   // tile granules (16 B) a thread stages with precomputed addresses
constexpr int ST_POS = 64;  // LDS bytes per halo-tile position (16-byte chunks XOR-swizzled by position)
constexpr int ST_STAGE = 32 * 144;  // per wave: 32 pixels x 128 output bytes (+16 pad), 1-byte outputs;
                                   // the four staging areas alias the input tile (dead during the 1x1 stage)

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
  int off_tile, off_pxoff, off_mid, off_cst, off_stage;  // LDS byte offsets (weight buffers at 0)
  int fast;             // 1: the fast requant path is valid (host proof, see conv_mfma.cuh store_group)
  int planes;           // 1: one 64-channel input chunk in LDS at a time; n_icc: all chunks resident (staged once per item)
  unsigned mg_g4, mg_lhw, mg_lw;  // ceil(2^32 / x) for x = 4 * planes, lh * lw, lw (resident-chunk staging)
  int occ_par;          // unfused only: 1 = a work item is (unit, output chunk) instead of a whole unit
#ifdef DFX_STAMPS
  unsigned long long *prof;  // diagnostic build only: [workgroup][wave][16] cycle sums
#endif
};

// The two fragment sets alternate per k-block; sched_barrier keeps hipcc from hoisting every
// fragment load of a stage to its top (register pressure) and from re-mixing the groups.
// (Round 1 read this fence as a workaround for a hardware write-after-read hazard on MFMA A/B
// operands.  There is none: deep-fusion_amd/tools/probe/probe_mfma_war.hip overwrote the operands
// 0..32 instructions after 2.6e10 MFMAs without one wrong result -- DESIGN.md section 4.1.)
#define DFX_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int OCC, int G, int PXB, int DST, bool FUSED>
__global__ __launch_bounds__(ST_THREADS, 2) void conv_stream_kernel(ConvArgs a, StreamGeom g) {
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  constexpr int WB = FUSED ? (OCC > G ? OCC : G) : OCC;  // fragments per half step a buffer holds
  constexpr int WBUF = 2 * WB * 1024;                    // bytes per weight buffer (three of them)
  constexpr int GA = 2 * OCC * 64, GB = 2 * G * 64;      // 16-byte granules per conv0 / conv1 step
  constexpr int NLD = (2 * WB * 64 + ST_THREADS - 1) / ST_THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const tile0 = smem + g.off_tile;
  unsigned char *tile = tile0;  // the chunk the current phase reads
  const int plane_bytes = g.npos * ST_POS;
  unsigned *pxoff = reinterpret_cast<unsigned *>(smem + g.off_pxoff);  // dst byte offset of each slot's pixel
  unsigned char *mid = smem + g.off_mid;
  float *cst0 = reinterpret_cast<float *>(smem + g.off_cst);  // FUSED: comp0 | bias0 | scale0 in LDS

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int OCP = 32 * g.ocb, OC1P = FUSED ? 32 * G * g.n_g1 : 0;
  const int *comp0 = reinterpret_cast<const int *>(FUSED ? cst0 : a.consts);
  const float *bias0 = (FUSED ? cst0 : a.consts) + OCP, *scale0 = (FUSED ? cst0 : a.consts) + 2 * OCP;
  const int *comp1 = reinterpret_cast<const int *>(a.consts + 3 * OCP);
  const float *bias1 = a.consts + 3 * OCP + OC1P, *scale1 = a.consts + 3 * OCP + 2 * OC1P;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int S0 = g.s0_steps, S1 = FUSED ? g.n_g1 * g.ks2 : 0, S = S0 + S1;
  const v4i *wsrc = reinterpret_cast<const v4i *>(a.wei);
  const int lane16 = lane * 16;
  const unsigned row_bytes = (unsigned)(FUSED ? a.oc1 : a.oc) * ESZ;

  // ---- weight stream: step t of a unit sits at granule woff(t) of the packed buffer ----
  v4i wreg[NLD];
  int wmsk = 0;  // granule mask of the step held in wreg
  const bool OCC_PAR = !FUSED && g.occ_par != 0;
  const int n_items = OCC_PAR ? g.total_units * g.n_occ : g.total_units;
  const int SC = OCC_PAR ? S0 / g.n_occ : S;  // steps per work item
  int f_tl = 0, f_item = blockIdx.x, f_base = OCC_PAR ? ((int)blockIdx.x % g.n_occ) * SC : 0;
// (No branch around a load or an LDS write: hipcc would wait vmcnt(0) inside each one.
// A step holds a power-of-two number of granules; surplus threads repeat a granule.)
// fetches the next step of this workgroup's step sequence into wreg and advances the
// sequence: the steps of a work item in order, then those of the item gridDim.x further on
#define DFX_W_FETCH()                                                                   \
  do {                                                                                  \
    const int t_ = f_base + f_tl;                                                       \
    const int off_ = t_ < S0 ? t_ * GA : S0 * GA + (t_ - S0) * GB;                      \
    wmsk = (t_ < S0 ? GA : GB) - 1;                                                     \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i)                                     \
      wreg[i] = wsrc[off_ + ((tid + ST_THREADS * i) & wmsk)];         \
    if (++f_tl == SC) {                                                                 \
      f_tl = 0;                                                                         \
      f_item += (int)gridDim.x;                                                         \
      f_base = (OCC_PAR && f_item < n_items) ? (f_item % g.n_occ) * SC : 0;             \
    }                                                                                   \
  } while (0)
#define DFX_W_COMMIT(BUF)                                                               \
  do {                                                                                  \
    v4i *d_ = reinterpret_cast<v4i *>(smem + (BUF) * WBUF);                             \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) d_[(tid + ST_THREADS * i) & wmsk] = wreg[i]; \
  } while (0)

  // ---- tile staging: granule q = tid + 256 i covers LDS position q >> 2, 16-byte
  //      chunk q & 3; its image / row / column inside the halo tile never change ----
  const int lhw = g.lh * g.lw;
  const int row_skip = g.lw - a.kw;  // positions from the last tap of a kernel row to the first of the next
  const int tile_q = g.npos * 4;
  int tq_pos[ST_TQ];  // img << 20 | ly << 10 | lx of the granule's position
#pragma unroll
  for (int i = 0; i < ST_TQ; ++i) {
    const int q = tid + ST_THREADS * i;
    const int pos = min(q >> 2, g.npos - 1);
    const int img = pos / lhw, r = pos - img * lhw;
    const int ly = r / g.lw, lx = r - ly * g.lw;
    tq_pos[i] = (img << 20) | (ly << 10) | lx;
  }
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};

  // ---- input tile prefetch registers (first ST_TQ granules per thread) ----
  v4i tv[ST_TQ];
  int tv_ok = 0;  // bit i: tv[i] is a real pixel granule (else padding)
#define DFX_T_ISSUE(N0, IY0, IX0, NIMG, ICC)                                            \
  do {                                                                                  \
    const int cb0_ = 64 * (ICC);                                                        \
    _Pragma("unroll") for (int i = 0; i < ST_TQ; ++i) {                                 \
      const int q_ = tid + ST_THREADS * i;                                              \
      const int img_ = tq_pos[i] >> 20, ly_ = (tq_pos[i] >> 10) & 1023, lx_ = tq_pos[i] & 1023; \
      const int iy_ = (IY0) + ly_, ix_ = (IX0) + lx_, cb_ = cb0_ + 16 * (q_ & 3);       \
      const bool ok_ = q_ < tile_q && img_ < (NIMG) && iy_ >= 0 && iy_ < a.ih && ix_ >= 0 && \
                       ix_ < a.iw && cb_ < a.ic;                                        \
      /* always an in-range address (clamped coordinates); padding is zeroed at commit: \
         no branch and no select around the load */                                     \
      const int n_ = min((N0) + img_, a.bs - 1), y_ = min(max(iy_, 0), a.ih - 1);       \
      const int x_ = min(max(ix_, 0), a.iw - 1), c_ = min(cb_, a.ic - 16);              \
      const long long o_ = (((long long)n_ * a.ih + y_) * a.iw + x_) * a.ic + c_;       \
      tv[i] = *reinterpret_cast<const v4i *>(a.src + o_);             \
      tv_ok = ok_ ? (tv_ok | (1 << i)) : (tv_ok & ~(1 << i));                           \
    }                                                                                   \
  } while (0)
/* granules beyond the tile go to a 16-byte dump slot right behind it */                  
#define DFX_T_COMMIT()                                                                  \
  do {                                                                                  \
    _Pragma("unroll") for (int i = 0; i < ST_TQ; ++i) {                                 \
      const int q_ = tid + ST_THREADS * i, pos_ = q_ >> 2;                              \
      const int lo_ = q_ < tile_q ? pos_ * ST_POS + 16 * ((q_ & 3) ^ chunk_swizzle<4>(pos_)) : g.npos * ST_POS; \
      *reinterpret_cast<v4i *>(tile + lo_) = ((tv_ok >> i) & 1) ? tv[i] ^ x80 : x80;    \
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
  const bool fast = g.fast != 0;
#ifdef DFX_STAMPS
  unsigned long long prof_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DFX_STAMP(t_entry);
  using TT = std::true_type;
  using FF = std::false_type;

  // Weight pipeline invariant at the start of step t: LDS buffers cur and cur+1 (mod 3)
  // hold steps t and t+1 (visible), wreg holds (in flight) step t+2.
  int cur = 0;
  DFX_W_FETCH();
  DFX_W_COMMIT(0);
  DFX_W_FETCH();
  DFX_W_COMMIT(1);
  DFX_W_FETCH();
  if (FUSED)
    for (int q = tid; q < 3 * OCP; q += ST_THREADS) cst0[q] = a.consts[q];  // visible after the first staging barrier
  bool tv_ready = false;  // tv holds the tile the next staging point needs
#define DFX_STEP_WEIGHTS()                                                              \
  {                                                                                     \
    int b2_ = cur + 2; if (b2_ >= 3) b2_ -= 3;                                          \
    DFX_W_COMMIT(b2_);                                                                  \
    DFX_W_FETCH();                                                                      \
  }
#define DFX_STEP_END()                 \
  __syncthreads();   \
  if (++cur == 3) cur = 0

  constexpr int NM0 = OCC * PXB, NM1 = G * PXB;  // MFMAs per k-block
  constexpr int NF0 = NM0 < 2 ? NM0 : 2, NF1 = NM1 < 2 ? NM1 : 2;
  // conv0 k-block fragments: set s of fb / fw
  v4i fb[2][PXB], fw[2][WB];
  // loads the fragments of the phase's next k-block (index kb) into set SET from weight buffer WBI, half J
#define DFX_LOAD0(SET, WBI, J)                                                          \
  do {                                                                                  \
    /* wave-uniform: this k-block's tap (in positions) and 32-channel half */            \
    const int tp_ = __builtin_amdgcn_readfirstlane(toff);                               \
    const int ch_ = __builtin_amdgcn_readfirstlane((kbn == 2 && (kb & 1)) ? 2 : 0);     \
    _Pragma("unroll") for (int pb = 0; pb < PXB; ++pb) {                                \
      const int P_ = fbase[pb] + tp_;                                                   \
      fb[SET][pb] = *reinterpret_cast<const v4i *>(tile + P_ * ST_POS + 16 * ((ch_ ^ h) ^ chunk_swizzle<4>(P_))); \
    }                                                                                   \
    const unsigned char *wb_ = smem + (WBI) * WBUF + (J) * OCC * 1024 + lane16;         \
    _Pragma("unroll") for (int r = 0; r < OCC; ++r)                                     \
      fw[SET][r] = *reinterpret_cast<const v4i *>(wb_ + r * 1024);                      \
    ++kb; /* a padding k-block (kb >= ns, zero weights) re-reads the last tap */        \
    if ((kbn == 1 || (kb & 1) == 0) && kb < ns) {                                       \
      toff += 1;                                                                        \
      if (++tkw == a.kw) { tkw = 0; toff += row_skip; }                                 \
    }                                                                                   \
  } while (0)
#define DFX_MFMA0(SET, M0, M1)                                                          \
  _Pragma("unroll") for (int m = (M0); m < (M1); ++m) {                                 \
    const int r = m / PXB, pb = m % PXB;                                                \
    acc[pb][r] = FUSED ? mfma_i8(fw[SET][r], fb[SET][pb], acc[pb][r])   /* D0[oc][px] */ \
                         : mfma_i8(fb[SET][pb], fw[SET][r], acc[pb][r]);  /* D0[px][oc] */ \
  }
#define DFX_LOAD1(SET, WBI, J)                                                          \
  do {                                                                                  \
    const int blk_ = min(kb, g.ocb - 1); /* a padding k-block has zero weights */       \
    _Pragma("unroll") for (int pb = 0; pb < PXB; ++pb)                                  \
      fb[SET][pb] = *reinterpret_cast<const v4i *>(my_mid[pb] + blk_ * 32);             \
    const unsigned char *wb_ = smem + (WBI) * WBUF + (J) * G * 1024 + lane16;           \
    _Pragma("unroll") for (int cc = 0; cc < G; ++cc)                                    \
      fw[SET][cc] = *reinterpret_cast<const v4i *>(wb_ + cc * 1024);                    \
    ++kb;                                                                               \
  } while (0)
#define DFX_MFMA1(SET, M0, M1)                                                          \
  _Pragma("unroll") for (int m = (M0); m < (M1); ++m) {                                 \
    const int cc = m / PXB, pb = m % PXB;                                               \
    acc1[pb][cc] = mfma_i8(fb[SET][pb], fw[SET][cc], acc1[pb][cc]);   \
  }

  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    DFX_STAMP(u0);
    const int unit = OCC_PAR ? item / g.n_occ : item;
    const int occ_lo = OCC_PAR ? item - unit * g.n_occ : 0, occ_hi = OCC_PAR ? occ_lo + 1 : g.n_occ;
    const int next_unit = OCC_PAR ? (item + (int)gridDim.x) / g.n_occ : item + (int)gridDim.x;
    const UnitGeo ug = unit_geo(unit);
    const int thc = min(g.thv, a.oh - ug.y0), twc = min(g.twv, a.ow - ug.x0);
    const int npx = ug.nimg * thc * twc;
    const bool has_next = item + (int)gridDim.x < n_items;
    const bool full = npx == ST_M * PXB;  // every slot holds a pixel: the stores need no predicate
    // this lane's pixel slots (conv0 column / conv1 row), one per pixel block
    int fbase[PXB];  // the slot's input position (tap 0) in the halo tile
    unsigned char *my_mid[PXB];
#pragma unroll
    for (int pb = 0; pb < PXB; ++pb) {
      const int slot = 32 * (wave * PXB + pb) + l31;
      const int pc = min(slot, npx - 1);
      const int img = pc / (thc * twc), r = pc - img * (thc * twc);
      const int ty = r / twc, tx = r - ty * twc;
      fbase[pb] = img * lhw + ty * a.sh * g.lw + tx * a.sw;
      if (h == 0)
        pxoff[slot] = slot < npx ? (unsigned)(((ug.n0 + img) * a.oh + ug.y0 + ty) * a.ow + ug.x0 + tx) * row_bytes
                                 : 0xffffffffu;
      my_mid[pb] = mid + slot * g.mid_stride + h * 16;
    }

    for (int occ = occ_lo; occ < occ_hi; ++occ) {
      v16i acc[PXB][OCC];
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int r = 0; r < OCC; ++r) acc[pb][r] = zero16;
      // unfused: this chunk's requant constants travel while the MFMAs run
      const int chb0 = 32 * OCC * occ + OCC * l31;  // unfused: lane owns channels chb0 .. chb0 + OCC-1
      int cp0[OCC];
      float bs0[OCC], sc0[OCC];
      if (!FUSED) {
#pragma unroll
        for (int cc = 0; cc < OCC; ++cc) {
          cp0[cc] = fast ? 0 : comp0[chb0 + cc];
          bs0[cc] = bias0[chb0 + cc];
          sc0[cc] = scale0[chb0 + cc];
        }
      }
      for (int icc = 0; icc < g.n_icc; ++icc) {
        // ---- stage input chunk icc of the halo tile (all waves are past the last
        //      step that read the previous contents: every step ends in a barrier).
        //      A single-chunk input stays in LDS for all output chunks of the unit. ----
        if (g.planes > 1) {
          // ---- all input chunks resident: stage them once per work item ----
          if (occ == occ_lo && icc == 0) {
            DFX_STAMP(s0);
            if (ESZ == 1 && (FUSED ? G == 4 : OCC >= 2)) __syncthreads();  // (store staging aliases the tile, see below)
            // granule q = position * (4 planes) + plane * 4 + chunk: consecutive threads read
            // a pixel's channels, then the next pixel's -- whole contiguous rows of src; the
            // index math uses host-made reciprocals, eight independent loads per pass
            const int g4 = 4 * g.planes, nq = g.npos * g4;
            auto fdiv = [](int q, int x, unsigned mg) { return x == 1 ? q : (int)__umulhi((unsigned)q, mg); };
            for (int q0 = tid; q0 < nq; q0 += 8 * ST_THREADS) {
              v4i v[8];
              int lo[8];
              bool ok[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int q = min(q0 + k * ST_THREADS, nq - 1);
                const int pos = fdiv(q, g4, g.mg_g4), c = q - pos * g4, pl = c >> 2, j = c & 3;
                const int img = fdiv(pos, lhw, g.mg_lhw), r = pos - img * lhw;
                const int ly = fdiv(r, g.lw, g.mg_lw), lx = r - ly * g.lw;
                const int iy = ug.iy0 + ly, ix = ug.ix0 + lx;
                ok[k] = img < ug.nimg && iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw && 64 * pl + 16 * j < a.ic;
                const int n_ = min(ug.n0 + img, a.bs - 1), y_ = min(max(iy, 0), a.ih - 1), x_ = min(max(ix, 0), a.iw - 1);
                const long long o = (((long long)n_ * a.ih + y_) * a.iw + x_) * a.ic + min(64 * pl + 16 * j, a.ic - 16);
                v[k] = *reinterpret_cast<const v4i *>(a.src + o);
                lo[k] = pl * plane_bytes + pos * ST_POS + 16 * (j ^ chunk_swizzle<4>(pos));
              }
#pragma unroll
              for (int k = 0; k < 8; ++k)  // (the clamped repeats at the end rewrite the last granule)
                *reinterpret_cast<v4i *>(tile0 + lo[k]) = ok[k] ? v[k] ^ x80 : x80;
            }
            __syncthreads();
            DFX_STAMP(s1);
            DFX_ACC(0, s1 - s0);
          }
          tile = tile0 + icc * plane_bytes;
        } else if (g.n_icc > 1 || occ == occ_lo) {
          DFX_STAMP(s0);
          // the 1-byte store staging area aliases the tile: every wave must be out of the
          // previous unit's epilogue before the new tile lands
          if (ESZ == 1 && (FUSED ? G == 4 : OCC >= 2) && occ == occ_lo && icc == 0) __syncthreads();
          if (!tv_ready) DFX_T_ISSUE(ug.n0, ug.iy0, ug.ix0, ug.nimg, icc);
          DFX_T_COMMIT();
          tv_ready = false;
          const int cb0 = 64 * icc;
          for (int q = tid + ST_THREADS * ST_TQ; q < tile_q; q += ST_THREADS) {  // oversized tiles
            const int pos = q >> 2, j = q & 3;
            const int img = pos / lhw, r = pos - img * lhw;
            const int ly = r / g.lw, lx = r - ly * g.lw;
            const int iy = ug.iy0 + ly, ix = ug.ix0 + lx;
            const bool ok = img < ug.nimg && iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw && cb0 + 16 * j < a.ic;
            const int n_ = min(ug.n0 + img, a.bs - 1), y_ = min(max(iy, 0), a.ih - 1), x_ = min(max(ix, 0), a.iw - 1);
            const long long o = (((long long)n_ * a.ih + y_) * a.iw + x_) * a.ic + min(cb0 + 16 * j, a.ic - 16);
            const v4i v = *reinterpret_cast<const v4i *>(a.src + o);
            *reinterpret_cast<v4i *>(tile + pos * ST_POS + 16 * (j ^ chunk_swizzle<4>(pos))) = ok ? v ^ x80 : x80;
          }
          __syncthreads();
          DFX_STAMP(s1);
          DFX_ACC(0, s1 - s0);
        }
        const int kbn = min(2, g.icb - 2 * icc);  // 32-channel blocks in this chunk
        const int ns = a.kh * a.kw * kbn, ns2 = (ns + 1) >> 1;  // k-blocks, steps
        int kb = 0, tkw = 0, toff = 0;  // next k-block to load, its tap column and tap offset in positions
        DFX_STAMP(p0);
        DFX_LOAD0(0, cur, 0);  // phase prologue (exposed): the first k-block's fragments
        DFX_FENCE();
        for (int s2 = 0; s2 < ns2; ++s2) {
          // k-block (s2, 0) from set 0, while set 1 <- (s2, 1)
          DFX_STAMP(x0);
          DFX_MFMA0(0, 0, NF0);
          DFX_FENCE();
          DFX_STAMP(x1);
          DFX_LOAD0(1, cur, 1);
          DFX_FENCE();
          DFX_STAMP(x2);
          DFX_MFMA0(0, NF0, NM0);
          DFX_FENCE();
          // k-block (s2, 1) from set 1, while set 0 <- (s2 + 1, 0) of the next buffer, and
          // the weight pipeline / tile prefetch move on
          DFX_STAMP(x3);
          DFX_MFMA0(1, 0, NF0);
          DFX_FENCE();
          DFX_STAMP(x4);
          if (s2 + 1 < ns2) {
            int nb_ = cur + 1; if (nb_ == 3) nb_ = 0;
            DFX_LOAD0(0, nb_, 0);
          }
          DFX_STAMP(x5);
          DFX_STEP_WEIGHTS();
          DFX_STAMP(x6);
          if (s2 == ns2 - 1) {  // last step before the next staging point: fetch its tile now
            if (g.planes > 1) {
              // resident chunks: nothing to prefetch here
            } else if (g.n_icc > 1 && (icc + 1 < g.n_icc || occ + 1 < occ_hi)) {
              DFX_T_ISSUE(ug.n0, ug.iy0, ug.ix0, ug.nimg, icc + 1 < g.n_icc ? icc + 1 : 0);
              tv_ready = true;
            } else if (!FUSED && occ + 1 == occ_hi && has_next) {
              const UnitGeo nx = unit_geo(next_unit);
              DFX_T_ISSUE(nx.n0, nx.iy0, nx.ix0, nx.nimg, 0);
              tv_ready = true;
            }
          }
          DFX_FENCE();
          DFX_STAMP(x7);
          DFX_MFMA0(1, NF0, NM0);
          DFX_FENCE();
          DFX_STAMP(x8);
          DFX_STEP_END();
          DFX_STAMP(x9);
          DFX_ACC(10, x1 - x0); DFX_ACC(11, x2 - x1); DFX_ACC(12, x3 - x2); DFX_ACC(13, x4 - x3);
          DFX_ACC(14, x5 - x4); DFX_ACC(15, x6 - x5); DFX_ACC(16, x7 - x6); DFX_ACC(17, x8 - x7);
          DFX_ACC(18, x9 - x8); DFX_ACC(19, 1);
        }
        DFX_STAMP(p1);
        DFX_ACC(1, p1 - p0);
      }
      DFX_STAMP(e0);
      if constexpr (FUSED) {
        // ---- requant 0 -> u8 -> this wave's rows of mid, in the 1x1 stage's k order:
        //      byte 16h + 4q + i of block r  =  channel 32r + 8q + 4h + i ----
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
        for (int r = 0; r < OCC; ++r) {
          v4i pkv;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ch = (occ * OCC + r) * 32 + 8 * q + 4 * h;
            const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
            const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
            unsigned pk = 0;
            if (fast) {  // bias slot = comp + bias (exact); ReLU + RNE + saturation in v_cvt_pk_u8_f32
#pragma unroll
              for (int i = 0; i < 4; ++i)  // plain v_add_f32 / v_mul_f32: the packed forms do not overlap with MFMAs (conv_mfma.cuh)
                pk = __builtin_amdgcn_cvt_pk_u8_f32(__fmul_rn(__fadd_rn(__int2float_rn(acc[pb][r][4 * q + i]), bs[i]), sc[i]), i, pk);
            } else {
              const v4i cp = *reinterpret_cast<const v4i *>(comp0 + ch);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float f = requant(acc[pb][r][4 * q + i] + cp[i], bs[i], sc[i], true);
                pk |= sat_u8_bits(cvt_x86_rt(f, a.rm0)) << (8 * i);
              }
            }
            pkv[q] = (int)(pk ^ 0x80808080u);
          }
          *reinterpret_cast<v4i *>(my_mid[pb] + (occ * OCC + r) * 32) = pkv;
        }
      } else {
        // ---- unfused: typed store; lane owns channels 32*OCC*occ + OCC*l31 + {0..OCC-1} ----
        if (chb0 < a.oc) {
          float zf[OCC];
#pragma unroll
          for (int cc = 0; cc < OCC; ++cc) zf[cc] = 0.0f;
          const bool relu = a.relu0 || DST == DFX_U8;
          unsigned char *dst_b = reinterpret_cast<unsigned char *>(a.dst);
          const unsigned chbE = (unsigned)chb0 * ESZ;
          auto emit0 = [&](auto fast_tag, auto check_tag) {
#pragma unroll
            for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
              for (int eq = 0; eq < 4; ++eq) {
                const v4i o4 = *reinterpret_cast<const v4i *>(pxoff + 32 * (wave * PXB + pb) + 8 * eq + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const unsigned off = (unsigned)o4[i];
                  if (!decltype(check_tag)::value || off != 0xffffffffu) {
                    int v[OCC];
#pragma unroll
                    for (int cc = 0; cc < OCC; ++cc) v[cc] = acc[pb][cc][4 * eq + i] + cp0[cc];
                    store_group<DST, OCC, decltype(fast_tag)::value>(dst_b + (size_t)(off + chbE), v, zf, bs0, sc0,
                                                                     relu, a.rm0);
                  }
                }
              }
          };
          if (ESZ == 1 && OCC >= 2 && occ + 1 == occ_hi) { /* staged below: every lane takes part */ }
          else if (fast) { if (full) emit0(TT{}, FF{}); else emit0(TT{}, TT{}); }
          else      { if (full) emit0(FF{}, FF{}); else emit0(FF{}, TT{}); }
        }
        if constexpr (!FUSED && ESZ == 1 && OCC >= 2) {
          // 1-byte outputs, the item's last output chunk (the tile is dead after its K loop:
          // every wave passed the last step's barrier): transpose 32 px x 32*OCC bytes through
          // LDS and write 16 bytes per lane instead of OCC bytes (see the fused stage)
          if (occ + 1 == occ_hi) {
            constexpr int RS = 32 * OCC + 16, C16 = 2 * OCC;  // staging row stride; 16-byte chunks per pixel
            unsigned char *stg = tile0 + wave * ST_STAGE;
            unsigned char *dst_b = reinterpret_cast<unsigned char *>(a.dst);
            const bool relu = a.relu0 || DST == DFX_U8;
            float zf[OCC];
#pragma unroll
            for (int cc = 0; cc < OCC; ++cc) zf[cc] = 0.0f;
            auto emit0s = [&](auto fast_tag) {
#pragma unroll
              for (int pb = 0; pb < PXB; ++pb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  int v[OCC];
#pragma unroll
                  for (int cc = 0; cc < OCC; ++cc) v[cc] = acc[pb][cc][e] + cp0[cc];
                  const unsigned pk = pack_group<DST, OCC, decltype(fast_tag)::value>(v, zf, bs0, sc0, relu, a.rm0);
                  unsigned char *w = stg + (8 * (e >> 2) + (e & 3) + 4 * h) * RS + OCC * l31;
                  if (OCC == 4) *reinterpret_cast<unsigned *>(w) = pk;
                  else *reinterpret_cast<unsigned short *>(w) = (unsigned short)pk;
                }
#pragma unroll
                for (int k = 0; k < C16 / 2; ++k) {
                  const int c = lane + 64 * k, px = c / C16, c16 = c % C16;
                  const unsigned off = pxoff[32 * (wave * PXB + pb) + px];
                  const v4i val = *reinterpret_cast<const v4i *>(stg + px * RS + 16 * c16);
                  const int ch = 32 * OCC * occ + 16 * c16;
                  if (off != 0xffffffffu && ch < a.oc) DFX_STORE16(reinterpret_cast<v4i *>(dst_b + (size_t)(off + ch)), val);
                }
              }
            };
            if (fast) emit0s(TT{}); else emit0s(FF{});
          }
        }
      }
      DFX_STAMP(e1);
      DFX_ACC(5, e1 - e0);
    }

    if constexpr (FUSED) {
      // ---- conv1 over mid, G column blocks at a time; the fragment pipeline runs through
      //      the groups (mid and every weight step of the unit are already in LDS) ----
      const bool relu = a.relu1 || DST == DFX_U8;
      int kb = 0;  // next k-block (= 32-channel block of mid) to load
      DFX_LOAD1(0, cur, 0);  // phase prologue
      DFX_FENCE();
      for (int g1 = 0; g1 < g.n_g1; ++g1) {
        DFX_STAMP(q0);
        v16i acc1[PXB][G];
#pragma unroll
        for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
          for (int cc = 0; cc < G; ++cc) acc1[pb][cc] = zero16;
        // this group's requant constants travel while the MFMAs run
        const int chb = 32 * G * g1 + G * l31;
        int cp[G];
        float bs[G], sc[G], zf[G];
#pragma unroll
        for (int cc = 0; cc < G; ++cc) {
          cp[cc] = fast ? 0 : comp1[chb + cc];
          bs[cc] = bias1[chb + cc];
          sc[cc] = scale1[chb + cc];
          zf[cc] = 0.0f;
        }
        for (int s2 = 0; s2 < g.ks2; ++s2) {
          DFX_MFMA1(0, 0, NF1);
          DFX_FENCE();
          DFX_LOAD1(1, cur, 1);
          DFX_FENCE();
          DFX_MFMA1(0, NF1, NM1);
          DFX_FENCE();
          DFX_MFMA1(1, 0, NF1);
          DFX_FENCE();
          if (s2 + 1 < g.ks2 || g1 + 1 < g.n_g1) {
            if (s2 + 1 == g.ks2) kb = 0;  // next group starts over on mid
            int nb_ = cur + 1; if (nb_ == 3) nb_ = 0;
            DFX_LOAD1(0, nb_, 0);
          }
          DFX_STEP_WEIGHTS();
          if (g.planes == 1 && g1 + 1 == g.n_g1 && s2 + 1 == g.ks2 && has_next) {  // next unit's first tile
            const UnitGeo nx = unit_geo(next_unit);
            DFX_T_ISSUE(nx.n0, nx.iy0, nx.ix0, nx.nimg, 0);
            tv_ready = true;
          }
          DFX_FENCE();
          DFX_MFMA1(1, NF1, NM1);
          DFX_FENCE();
          DFX_STEP_END();
        }
        DFX_STAMP(e2);
        DFX_ACC(2, e2 - q0);
        if ((ESZ == 1 && G == 4) || chb < a.oc1) {
          unsigned char *dst_b = reinterpret_cast<unsigned char *>(a.dst);
          const unsigned chbE = (unsigned)chb * ESZ;
          auto emit1 = [&](auto fast_tag, auto check_tag) {
#pragma unroll
            for (int pb = 0; pb < PXB; ++pb)
#pragma unroll
              for (int eq = 0; eq < 4; ++eq) {
                const v4i o4 = *reinterpret_cast<const v4i *>(pxoff + 32 * (wave * PXB + pb) + 8 * eq + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const unsigned off = (unsigned)o4[i];
                  if (!decltype(check_tag)::value || off != 0xffffffffu) {
                    int v[G];
#pragma unroll
                    for (int cc = 0; cc < G; ++cc) v[cc] = acc1[pb][cc][4 * eq + i] + cp[cc];
                    store_group<DST, G, decltype(fast_tag)::value>(dst_b + (size_t)(off + chbE), v, zf, bs, sc, relu,
                                                                   a.rm1);
                  }
                }
              }
          };
          // 1-byte outputs with 4 column blocks: a dword store per pixel costs as much issue
          // time as a 16-byte one, so the wave transposes its 32 x 128 bytes through LDS
          // and writes 16 bytes per lane (4 stores per pixel block instead of 16)
          auto emit1s = [&](auto fast_tag) {
            if constexpr (ESZ == 1 && G == 4) {
              unsigned char *stg = smem + g.off_stage + wave * ST_STAGE;
#pragma unroll
              for (int pb = 0; pb < PXB; ++pb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  int v[G];
#pragma unroll
                  for (int cc = 0; cc < G; ++cc) v[cc] = acc1[pb][cc][e] + cp[cc];
                  const unsigned pk = pack_group<DST, G, decltype(fast_tag)::value>(v, zf, bs, sc, relu, a.rm1);
                  *reinterpret_cast<unsigned *>(stg + (8 * (e >> 2) + (e & 3) + 4 * h) * 144 + 4 * l31) = pk;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int c = lane + 64 * k, px = c >> 3, c16 = c & 7;
                  const unsigned off = pxoff[32 * (wave * PXB + pb) + px];
                  const v4i val = *reinterpret_cast<const v4i *>(stg + px * 144 + 16 * c16);
                  if (off != 0xffffffffu && 128 * g1 + 16 * c16 < a.oc1)
                    DFX_STORE16(reinterpret_cast<v4i *>(dst_b + (size_t)(off + 128 * g1 + 16 * c16)), val);
                }
              }
            }
          };
          if (ESZ == 1 && G == 4) { if (fast) emit1s(TT{}); else emit1s(FF{}); }
          else if (fast) { if (full) emit1(TT{}, FF{}); else emit1(TT{}, TT{}); }
          else      { if (full) emit1(FF{}, FF{}); else emit1(FF{}, TT{}); }
        }
        DFX_STAMP(e3);
        DFX_ACC(6, e3 - e2);
      }
    }
    DFX_STAMP(u1);
    DFX_ACC(7, u1 - u0);
    DFX_ACC(8, 1);
  }
#ifdef DFX_STAMPS
  {
    DFX_STAMP(t_end);
    if (lane == 0) {
      unsigned long long *o = g.prof + ((size_t)blockIdx.x * 4 + wave) * 24;
      for (int k = 0; k < 24; ++k) o[k] = prof_acc[k];
      o[9] = t_end - t_entry;
    }
  }
#endif
#undef DFX_W_FETCH
#undef DFX_W_COMMIT
#undef DFX_T_ISSUE
#undef DFX_T_COMMIT
#undef DFX_STEP_WEIGHTS
#undef DFX_STEP_END
#undef DFX_LOAD0
#undef DFX_MFMA0
#undef DFX_LOAD1
#undef DFX_MFMA1
}

}  // namespace dfx
