// conv_direct.cuh -- fused u8 x s8 conv (+ReLU) + conv1x1 (+ReLU) for general shapes as two
// int8-MFMA implicit GEMMs whose WEIGHT fragments go straight from global memory (L2) into
// the MFMA operand registers of the one wave that needs them (gfx950 / CDNA4).
//
// Same contract as conv_stream.cuh (/root/reference/src/jit_conv_kernel.cc:143-393 with the
// multi-chunk accumulation of :193-216, any kernel size / stride / padding, channels
// multiples of 16), different decomposition.  conv_stream.cuh gives every wave 32 pixels and
// ALL output channels, so the four waves share each weight fragment through LDS: a staging
// copy and a workgroup barrier per 8 MFMAs.  Here a workgroup of NW = 4 or 8 waves owns a unit
// of NPB = 1, 2 or 4 blocks of 32 output pixels, and
//  * in conv0 the waves split the output-channel blocks WO ways and the pixel blocks WP = NW / WO ways: wave
//    (wo, wp) owns blocks wo, wo + WO, ... for its PXW = NPB / WP pixel blocks.  Its weight fragments are
//    private: ONE stream of 1 KB blocks (k-blocks 0 .. nkb0 - 1 of each of its output blocks) through a ring of
//    DK_RD register quadruples, buffer_load_dwordx4 with a scalar offset, fed to the MFMA as the A operand.
//    No weight staging, no barrier inside the K loop.
//  * the input halo tile is staged ONCE per unit with all its 64-channel planes ([plane][image][row][col][80 B];
//    80 = 5 x 16 spreads consecutive columns over the bank columns, and the host pads the row / image pitch by
//    16 c bytes so that a 32-pixel block that wraps over several output rows still reads without bank conflicts:
//    DirectGeom::row_pitch), read-only afterwards; a fragment address is base + a wave-uniform offset;
//  * the u8 intermediate goes to LDS (mid[slot][oc], the reference keeps it in xmm registers,
//    jit_conv_kernel.cc:275-277); one barrier; then conv1 with the same idea: WO1 x WP1 waves, wave (wo1, wp1)
//    owns 1x1 output groups wo1, wo1 + WO1, ... (G column blocks each, the channel permutation of
//    conv_mfma.cuh) for its PXW1 pixel blocks, PX1 of them per pass; A fragments from mid, B fragments through
//    a second ring that is primed before the barrier and runs on across passes, groups and store epilogues.
// Three workgroup barriers per unit.
//
// What round 3 changed, each with its measurement (N = 128, u8 out; profiles/r03/):
//  * The K loops are BRANCH-FREE around their loads.  hipcc's wait-count pass merges the states of the two sides
//    of a branch, so with the ring refill inside `if (kb < nkb0)` it counted the ring down to vmcnt(0) once per
//    DK_RD k-blocks: the "9-deep" ring was drained every round (85 cycles per MFMA at res4; direct_sweep_1 /
//    stamps_direct_1 before, _3 / _4 after).  Whole rounds now run unconditionally (EVEN: nkb0 a multiple of
//    DK_RD, every 3x3 layer -- refills run on into the next output block and the requant of one block hides
//    under the loads of the next; otherwise the last k-blocks of a block run unpipelined).  conv1's ring depth
//    divides the (padded) block count, so that loop has no odd tail at all.
//  * Weights come through buffer loads (scalar offset + one lane-offset VGPR).  With flat addresses hipcc hoisted a
//    64-bit per-lane pointer per ring slot out of the unit loop and spilled 70-84 VGPRs (~20 k cycles of scratch
//    reloads per unit).  Both stages' constants live in LDS (LDS-DMA at kernel entry): a global load in the store
//    epilogue waited for the ring.
//  * NW = 8 (one workgroup per CU, two waves per SIMD) for layers whose units do not fill the CUs twice: res4
//    39 -> 35 us, res5 58 -> 45 us (and one launch instead of round 2's two: 79 us).
//  * The requant routes are template parameters for u8 output (QM: generic / fma + magic / fma + fma).  As
//    run-time branches around the store epilogue they made hipcc copy every accumulator out of its MFMA tuple
//    (64 v_mov per 32 x 128 block, 256 VGPRs, spills).  The specialised kernels fit PX1 = 2 pixel blocks per conv1
//    pass (W1 crosses the L2 port half as often) and store dwords straight from the accumulators.
//  * Requant without int -> float conversions where the host proves the ranges (dfx_api.hip, as for
//    conv_mfma_roles.cuh): stage 0 "fma" (accumulators start from bits(2^23) + comp + bias, one v_fma_f32 +
//    v_cvt_pk_u8_f32 per value), stage 1 "magic" (start 1/(2 pi), v_add_f32 + v_mul_f32 + cvt) or "fma".
//  * Bank-conflict-free tile pitches (above): 47 % of the LDS cycles at res4 were conflicts; res4 30.2 -> 28.2 us.
//  * Set-up arithmetic: f32-reciprocal quotients instead of integer division sequences in the staging table and
//    the per-unit slot decode (~350 instructions per unit).
// Tried and dropped (A/B logs in profiles/r03/ab_direct_*.txt): starting the second half of the grid 6-13 k cycles late
// to de-phase the two workgroups of a CU (slower by the delay), priming the conv0 ring before the tile is staged,
// conv1 ring depth 4 with PX1 = 2, tile loads one unit ahead, and RASTER units (128 consecutive pixels of the batch
// across row and image ends, so that every unit is full: 784 instead of 896 units at 28 x 28) -- bit-exact on all
// shapes, but the taller tile (10 rows instead of 6), its LDS and the per-unit segment table's scalars cost more
// than the full units won: res3 44.9 vs 38.5 us on one box, and the extra live scalars slowed the rectangular
// path by 8-15 % as well (this kernel sits at the SGPR limit: ~100 spilled to VGPR lanes).
//  * 3x3 fast path (T9 below): no scalar bookkeeping in the conv0 K loop -- the generic loop's ~30 scalar
//    instructions per k-block made res5 (one MFMA per k-block) scalar-issue bound.  res5 43.1 -> 38.1 us, res4
//    29.9 -> 27.4, res3 35.3 -> 33.1.
//  * vector-instruction diet (late round 3; PMC: res3 issued 8.4 M non-MFMA vector instructions per launch where the
//    two requants need 2.0 M): accumulator start values as the C operand of each chain's first MFMA (an inline
//    constant for conv1: 16 v_mov per accumulator and pass gone), the tile's granules beyond the first 6 per thread
//    in batches of 2 (res3's 144 left-over granules cost every thread a whole batch of 6), the conv0 fragment bases
//    from the slot table instead of two divisions per pixel block: res3 33.5 -> 32.1 us, res3s2 41.8 -> 39.4,
//    vgg3 124.0 -> 117.6, res4 27.8 -> 27.2 (profiles/r03/ab_direct_valu_trims.txt).  Tried on top and dropped:
//    the tile's loads as buffer loads with 32-bit offsets relative to the unit's first image and no coordinate
//    clamps (2-7 % SLOWER, same file).
//    And: stage 1's add + mul route ("magic", taken where the scale is not a power of two: res3, res5) as ONE fma from
//    per-channel accumulator start values -- bits(2^23) + comp + bias as in stage 0, brought in as 16-register tuples
//    by four ds_read_b128 of an LDS table entry {v, v, v, v} per accumulator: 256 fewer vector instructions per wave
//    and unit, 64 more LDS reads, bit-exact, and 1-2 % SLOWER (res3 33.2 -> 33.9 us, res5 38.1 -> 38.6).
// Where the time goes now (stamps build, profiles/r03/stamps_direct_12_after_fast_path.txt): res4 conv0 K loop 38 %
// of a unit (the matrix pipe is ~100 % busy inside it), conv1 23 % (77 %), tile staging 11 %, store epilogue 9.5 %,
// barrier imbalance 8 %; ~5.6 k cycles (10 %) from kernel entry to the first unit.  The weight stream is the floor
// of res5: 3.4 MB per 32-pixel unit through one CU's L2 port at ~60 B/clk.
#pragma once

#include "conv_mfma.cuh"
#include "conv_pw.cuh"  // pw_quarter: one quarter block after requant, as values

namespace dfx {

constexpr int DK_M = 128;    // pixel slots per unit at most (NPB = 4 blocks of 32)
constexpr int DK_POS = 80;   // LDS bytes per halo-tile position and plane (64 + 16 pad)
constexpr int DK_TQ = 6;     // tile granules a thread holds in registers at once (4 waves: 24 KB of tile, 8 waves: 48 KB)
constexpr int DK_RD = 9;     // conv0 weight ring: k-blocks in flight per wave (multiple of 3)
constexpr int DK_STAGE = 32 * 144;  // per wave: 1-byte store staging (aliases the dead tile)
#ifndef DK_QM_TRANSPOSE
#define DK_QM_TRANSPOSE 0           // 1: the conversion-free routes store through the LDS transpose too (A/B aid)
#endif

struct DirectGeom {
  int ni, thv, twv;     // unit = ni whole images (ni > 1 only if thv == oh && twv == ow) x thv x twv px
  int uy, ux, total_units;
  int lh, lw, npos;     // halo tile rows / cols per image; positions = ni * lh * lw
  int icb;              // 32-channel input blocks
  int n_planes;         // 64-channel planes of the tile = (icb + 1) / 2
  int plane_bytes;      // ni * img_pitch
  // byte pitches of the tile: position (img, ly, lx) of a plane sits at img * img_pitch + ly * row_pitch + lx *
  // DK_POS.  row_pitch = lw * DK_POS + 16 c, img_pitch = lh * row_pitch + 16 c': the host picks c, c' (0..15) so
  // that the 16 lanes of every ds_read_b128 lane group of a pixel fragment fall into 16 different 16-byte bank
  // columns.  (With the plain pitches consecutive slots of one output row are conflict-free -- DK_POS is 5 x 16 --
  // but a 32-slot block wraps over 2-5 output rows and the row gap broke the pattern: 47 % of the LDS cycles at
  // res4 were bank conflicts, profiles/r03/final pmc.)
  int row_pitch, img_pitch;
  // unfused != 0: no 1x1 stage (oc1x1 == 0).  conv0's requantised result IS the output: 4-byte outputs are stored
  // straight from the accumulators (a lane holds 4 consecutive channels of its pixel per q: one 16-byte store),
  // 1-byte outputs are collected in `mid` in natural channel order and leave as whole pixel rows (16-byte stores).
  int unfused;
  int ocb;              // conv0 output blocks, padded to a multiple of WO
  int n_g1;             // conv1 groups of G column blocks
  int mid_stride;       // 32 * ocb + 16
  int off_pxoff, off_mid, off_cst;  // LDS byte offsets (tile and the aliased staging at 0)
  int fast;             // 1: fast requant path valid (host proof)
  int m0, m1;           // host-proven requant without conversions: m0 = 1 stage-0 "fma"; m1 = 2 / 3 stage-1 "magic" / "fma"
  int npb;              // 32-pixel blocks per unit (1, 2 or 4)
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

// NW waves per workgroup: 4 (two workgroups per CU where the units and the LDS allow) or 8 (one workgroup per CU,
// two waves per SIMD: layers whose 128-pixel units do not even fill the CUs once)
// QM: requant routes fixed at compile time.  0: both stages choose exact / fast (/ stage-0 fma) at run time;
// 1 / 2 (u8 output through the 16-byte store path): stage 0 "fma", stage 1 "magic" / "fma" -- with the four
// routes of the store epilogue as run-time branches hipcc copied every accumulator out of its MFMA tuple at the
// head of the taken branch (64 v_mov per 32 x 128 block and twice the registers: 256 VGPRs and spills).
template <int NW, int WO, int G, int WO1, int DST, int NPB, int QM>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_direct_kernel(ConvArgs a, DirectGeom g) {
  static_assert(QM == 0 || (DST == DFX_U8 && G == 4), "the conversion-free routes exist for u8 output with G = 4");
  constexpr int DK_THREADS = 64 * NW;
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  constexpr int WP = NW / WO, PXW = NPB / WP;        // conv0: WO x WP waves, PXW pixel blocks per wave
  constexpr int WP1 = NW / WO1, PXW1 = NPB / WP1;    // conv1 likewise
  static_assert(WO * WP == NW && WO1 * WP1 == NW && WP * PXW == NPB && WP1 * PXW1 == NPB && PXW >= 1 && PXW1 >= 1,
                "wave split must tile the unit's pixel blocks");
  // pixel blocks per conv1 pass (accumulators: PX1 * G * 16 VGPRs)
  constexpr int PX1 = G == 4 ? ((QM != 0 && PXW1 >= 2) ? 2 : 1) : (PXW1 > 2 ? 2 : PXW1);
  constexpr int NP1 = PXW1 / PX1;             // conv1 passes per group
  // conv1 weight ring depth in k-blocks (G fragments each).  Even, and it divides ocb (a multiple of WO): the
  // ring never has to stop at the end of a group.
  constexpr int RD1 = (WO % 4 != 0 || PX1 * G > 4) ? 2 : 4;  // (4 with PX1 * G = 8: 256 VGPRs and spills, 3-6 % slower)
  static_assert(WO % RD1 == 0, "the conv1 ring must divide the number of output blocks");
  constexpr int NF = PXW < 2 ? PXW : 2;       // MFMAs issued before the k-block's LDS prefetch
  DFX_STAMP(t_entry);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const tile0 = smem;
  unsigned *pxoff = reinterpret_cast<unsigned *>(smem + g.off_pxoff);  // [32 NPB] dst byte offset of the slot's pixel
  int *fboff = reinterpret_cast<int *>(pxoff + 32 * NPB);                // [32 NPB] tile byte offset of its input position
  unsigned char *mid = smem + g.off_mid;
  float *cst0 = reinterpret_cast<float *>(smem + g.off_cst);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wo = wave / WP, wp = wave % WP, wo1 = wave / WP1, wp1 = wave % WP1;
  const int OCP = 32 * g.ocb, OC1P = 32 * G * g.n_g1;
  const int *comp0 = reinterpret_cast<const int *>(cst0);
  const float *bias0 = cst0 + OCP, *scale0 = cst0 + 2 * OCP;
  const int *comp1 = reinterpret_cast<const int *>(cst0 + 3 * OCP);  // (LDS as well: a global load in the store epilogue would wait for the weight ring)
  const float *bias1 = cst0 + 3 * OCP + OC1P, *scale1 = cst0 + 3 * OCP + 2 * OC1P;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int ntap = a.kh * a.kw, nkb0 = g.icb * ntap;
  // weights: a.wei = W0d[ocb][nkb0][64 lanes][16 B], a.wei1 = W1d[n_g1][ocb][G][64 lanes][16 B]
  const unsigned row_bytes = (unsigned)(g.unfused ? a.oc : a.oc1) * ESZ;
  const bool fast = g.fast != 0;
  const bool relu1 = a.relu1 || DST == DFX_U8;
  using TT = std::true_type;
  using FF = std::false_type;

  // Both stages' constants -> LDS by LDS-DMA (a wave moves 1 KB per instruction; no registers, nothing to wait for
  // here): they land while the first unit's tile is being staged (vector memory operations complete in order, and
  // the first unit's second barrier is preceded by s_waitcnt vmcnt(0)).  Through registers this copy cost 3-5 k
  // cycles at kernel entry (6-9 k from entry to the first unit: profiles/r03/stamps_direct_7_startup.txt).
  {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void global_void;
    const int total16 = 3 * (OCP + OC1P) / 4;  // 16-byte chunks (OCP, OC1P: multiples of 32)
    const v4i *cs = reinterpret_cast<const v4i *>(a.consts);
    v4i *cd = reinterpret_cast<v4i *>(cst0);
    for (int j = wave; 64 * j < total16; j += NW) {
      const int q = 64 * j + lane;
      if (q < total16) __builtin_amdgcn_global_load_lds((global_void *)(cs + q), (lds_void *)(cd + 64 * j), 16, 0, 0);
    }
  }
  // ---- tile staging: granule q = tid + 256 i -> plane q / (4 npos), position, 16-byte chunk q & 3 ----
  const int lhw = g.lh * g.lw;
  const int row_skip = g.row_pitch - a.kw * DK_POS;  // bytes from the last tap of a kernel row to the next row's first
  const int tile_q1 = g.npos * 4, tile_q = tile_q1 * g.n_planes;
  int tq_pos[DK_TQ];  // plane << 28 | img << 20 | ly << 10 | lx
  // (quotients through f32 reciprocals with a +-1 fix-up: every dividend is < 2^22; the integer division sequence
  // is ~35 instructions and there are 3 per entry, all on the path from kernel entry to the first tile load)
  auto divmod = [](int x, int d, float rd, int &rem) -> int {
    int q = (int)(__int2float_rn(x) * rd);
    rem = x - q * d;
    if (rem < 0) { --q; rem += d; }
    if (rem >= d) { ++q; rem -= d; }
    return q;
  };
  const float r_tq1 = 1.0f / __int2float_rn(tile_q1), r_lhw = 1.0f / __int2float_rn(lhw), r_lw = 1.0f / __int2float_rn(g.lw);
  auto tq_fill = [&](auto &tqp, int qbase) {
    constexpr int N_ = (int)(sizeof(tqp) / sizeof(int));
#pragma unroll
    for (int i = 0; i < N_; ++i) {
      const int q = min(qbase + tid + DK_THREADS * i, tile_q - 1);
      int ql, r, lx;
      const int pl = divmod(q, tile_q1, r_tq1, ql);
      const int img = divmod(ql >> 2, lhw, r_lhw, r);
      const int ly = divmod(r, g.lw, r_lw, lx);
      tqp[i] = (pl << 28) | (img << 20) | (ly << 10) | lx;
    }
  };
  tq_fill(tq_pos, 0);
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
  v4i tv[DK_TQ];
  int tv_ok = 0;
  // (branch-free loads: hipcc waits vmcnt(0) inside a branch around a load)
#define DK_T_ISSUE(TV, TQP, QB, N0, IY0, IX0, NIMG)                                     \
  do {                                                                                  \
    /* scalar base of the unit's first image + a 32-bit per-lane offset (the host admits only units whose images  \
       span < 2^31 bytes): no 64-bit per-lane multiply-adds */                                                   \
    const long long img_b_ = (long long)a.ih * a.iw * a.ic;                             \
    const uint8_t *ub_ = a.src + (long long)(N0) * img_b_;                              \
    const int nmax_ = a.bs - 1 - (N0);                                                  \
    _Pragma("unroll") for (int i = 0; i < (int)(sizeof(TQP) / sizeof(int)); ++i) {      \
      const int q_ = (QB) + tid + DK_THREADS * i;                                       \
      const int pl_ = (TQP[i] >> 28) & 15, img_ = (TQP[i] >> 20) & 255;                 \
      const int ly_ = (TQP[i] >> 10) & 1023, lx_ = TQP[i] & 1023;                       \
      const int iy_ = (IY0) + ly_, ix_ = (IX0) + lx_, cb_ = 64 * pl_ + 16 * (q_ & 3);   \
      const bool ok_ = q_ < tile_q && img_ < (NIMG) && iy_ >= 0 && iy_ < a.ih && ix_ >= 0 && \
                       ix_ < a.iw && cb_ < a.ic;                                        \
      /* always an in-range address (clamped coordinates); padding is zeroed at commit */ \
      const int n_ = min(img_, nmax_), y_ = min(max(iy_, 0), a.ih - 1);                 \
      const int x_ = min(max(ix_, 0), a.iw - 1), c_ = min(cb_, a.ic - 16);              \
      const unsigned o_ = (unsigned)(((n_ * a.ih + y_) * a.iw + x_) * a.ic + c_);       \
      (void)DK_CHK(1, (long long)(N0) * img_b_ + o_, 16, g.src_bytes);                  \
      TV[i] = *reinterpret_cast<const v4i *>(ub_ + o_);                                 \
      tv_ok = ok_ ? (tv_ok | (1 << i)) : (tv_ok & ~(1 << i));                           \
    }                                                                                   \
  } while (0)
#define DK_T_COMMIT(TV, TQP, QB)                                                        \
  do {                                                                                  \
    _Pragma("unroll") for (int i = 0; i < (int)(sizeof(TQP) / sizeof(int)); ++i) {      \
      const int q_ = (QB) + tid + DK_THREADS * i;                                       \
      const int pl_ = (TQP[i] >> 28) & 15, img_ = (TQP[i] >> 20) & 255;                 \
      const int ly_ = (TQP[i] >> 10) & 1023, lx_ = TQP[i] & 1023;                       \
      const int lo_ = q_ < tile_q ? pl_ * g.plane_bytes + img_ * g.img_pitch + ly_ * g.row_pitch + lx_ * DK_POS + 16 * (q_ & 3) \
                                  : g.n_planes * g.plane_bytes; /* dump slot */          \
      *reinterpret_cast<v4i *>(tile0 + lo_) = ((tv_ok >> i) & 1) ? TV[i] ^ x80 : x80;   \
    }                                                                                   \
  } while (0)

  // buffer resources of the two weight arrays (raw, range-checked: W0d ocb * nkb0 KB, W1d n_g1 * ocb * G KB).
  // With flat addresses hipcc hoisted one 64-bit per-lane pointer per ring slot out of the unit loop and
  // spilled them (70-84 VGPRs, ~20 k cycles of scratch reloads per unit).
  const __amdgpu_buffer_rsrc_t w0rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(a.wei), 0, g.ocb * nkb0 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(a.wei1), 0, g.n_g1 * g.ocb * G * 1024, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int ob_last = wo + (g.ocb - 1 - wo) / WO * WO;
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

  bool first_unit = true;
#ifdef DFX_STAMPS
  unsigned long long prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool prof_first = true;
#endif

  // (Tried and dropped in round 3: issuing a unit's tile loads one unit ahead, after the last K loop of the unit
  // before it -- 3-6 % SLOWER at res3 / res4, profiles/r03/ab_direct_tile_ahead.txt: the loads return in order, so
  // they cannot go out earlier than that without stalling the weight rings, and the 16-24 extra live registers
  // cost more than the one store epilogue they overlap.)
  for (int unit = blockIdx.x; unit < g.total_units; unit += gridDim.x) {
    const UnitGeo ug = unit_geo(unit);
    // (keeps the staging table's decoded fields and addresses from being hoisted out of the unit loop: as loop
    // invariants they outlived both K loops and were spilled)
#pragma unroll
    for (int i = 0; i < DK_TQ; ++i) asm volatile("" : "+v"(tq_pos[i]));
    const int thc = min(g.thv, a.oh - ug.y0), twc = min(g.twv, a.ow - ug.x0);
    const int npx = ug.nimg * thc * twc;
    DFX_STAMP(t0);
#ifdef DFX_STAMPS
    if (prof_first) { prof_acc[8] = t0 - t_entry; prof_first = false; }  // kernel entry -> first unit
#endif
    // every wave is out of the previous unit (its store staging aliases the tile; pxoff, mid)
    __syncthreads();
    // the conv0 weight ring (see below).  (Priming it here, before the tile is staged, bought nothing: +-2 %,
    // profiles/r03/ab_direct_early_prime.txt, for 30 more live VGPRs.)
    int r_ob = wo, r_kb = 0;  // next block the ring fetches
    auto r_next = [&]() -> unsigned {  // its index in W0d, then advance (past the stream's end: the last output block again)
      const unsigned idx = (unsigned)(min(r_ob, ob_last) * nkb0 + r_kb);
      const bool wrap = r_kb + 1 == nkb0;
      r_kb = wrap ? 0 : r_kb + 1;
      r_ob = wrap ? r_ob + WO : r_ob;
      return idx;
    };
    auto wload = [&](unsigned idx) -> v4i {  // buffer_load_dwordx4 v, lane16, rsrc, soffset: no per-lane 64-bit addresses
      return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(w0rs, (int)lane16, (int)(idx << 10), 0));
    };
    v4i wr[DK_RD];
    const bool even0 = nkb0 % DK_RD == 0;
    // ---- stage the whole halo tile (all planes) ----
    DK_T_ISSUE(tv, tq_pos, 0, ug.n0, ug.iy0, ug.ix0, ug.nimg);
    // (slot -> image, row, column of the unit: f32-reciprocal quotients as in the staging table; the integer
    // division sequences of the slot table and of conv0's PXW fragment bases were ~350 instructions per unit)
    const float r_px = 1.0f / __int2float_rn(thc * twc), r_tw = 1.0f / __int2float_rn(twc);
    if (wave < NPB) {  // slot table: wave w fills pixel block w
      const int slot = 32 * wave + l31;
      const int pc = min(slot, npx - 1);
      int r, tx;
      const int img = divmod(pc, thc * twc, r_px, r);
      const int ty = divmod(r, twc, r_tw, tx);
      // (pc is clamped to the unit's last pixel.  QM != 0: empty slots repeat that pixel's offset -- see the store
      // epilogue; otherwise they are marked)
      if (h == 0) {
        pxoff[slot] = (slot < npx || (QM != 0 && !DK_QM_TRANSPOSE)) ? (unsigned)(((ug.n0 + img) * a.oh + ug.y0 + ty) * a.ow + ug.x0 + tx) * row_bytes
                                                                   : 0xffffffffu;
        // (the conv0 waves used to redo the slot's two divisions for each of their PXW pixel blocks)
        fboff[slot] = img * g.img_pitch + ty * a.sh * g.row_pitch + tx * a.sw * DK_POS;
      }
    }
    DK_T_COMMIT(tv, tq_pos, 0);
    // the part of the tile beyond the first DK_TQ granules per thread (stride-2 tiles, many planes): further batches
    // of DK_TQ2 loads each, positions computed on the fly.  (Until late round 3 this was a loop of single dependent
    // load -> store pairs with three integer divisions each: 744 of stride-2 res3's 2280 granules went through it.
    // Then batches of DK_TQ = 6: res3's 1680-granule tile has 144 granules beyond the first 1536, and all 256 threads
    // walked a whole second batch for them -- ~570 vector instructions per wave and unit, about a fifth of all it
    // issued; with batches of 2 it is a third of that.)
    constexpr int DK_TQ2 = 2;
    for (int qb = DK_THREADS * DK_TQ; qb < tile_q; qb += DK_THREADS * DK_TQ2) {
      int tq2[DK_TQ2];
      v4i tv2[DK_TQ2];
      tq_fill(tq2, qb);
      DK_T_ISSUE(tv2, tq2, qb, ug.n0, ug.iy0, ug.ix0, ug.nimg);
      DK_T_COMMIT(tv2, tq2, qb);
    }
    if (first_unit) {  // the constants' LDS-DMA (see above)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      first_unit = false;
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
      fbyte[p] = fboff[slot] + 16 * h;
      mid_w[p] = mid + slot * g.mid_stride + h * 16;
    }
    // requant 0 -> u8 -> mid, in the 1x1 stage's k order: byte 16h + 4q + i of block ob = channel 32 ob + 8q + 4h + i.
    // MODE 2: "fma" (bias slot = -2^23 * scale, see conv_mfma_roles.cuh), 1: fast, 0: exact
    auto requant0 = [&](auto mode_tag, int ob, const v16i(&acc)[PXW]) {
      constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
      for (int p = 0; p < PXW; ++p) {
        v4i pkv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = ob * 32 + 8 * q + 4 * h;
          const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
          const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
          unsigned pk = 0;
          if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(__int_as_float(acc[p][4 * q + i]), sc[i], bs[i]), i, pk);
          } else if constexpr (MODE == 1) {
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
    };
    // The weight fragments of this wave are ONE stream of 1 KB blocks (output blocks wo, wo + WO, ...; k-blocks
    // 0 .. nkb0 - 1 of each) fetched through a ring of DK_RD registers quadruples.  The K loop is branch-free
    // around its loads: hipcc's wait-count pass merges the states of the two sides of a branch, so a load inside
    // a conditional made it count the ring down to vmcnt(0) once per DK_RD k-blocks (round 2 / early round 3:
    // the ring was drained every 9 k-blocks, 85 cycles per MFMA at res4).  EVEN (nkb0 a multiple of DK_RD, every
    // 3x3 layer): refills run on into the next output block, the requant of one block hides under the loads of
    // the next.  Otherwise whole rounds are branch-free and the last k-blocks of a block run unpipelined.
    // unfused op: requant 0 IS the output stage (see DirectGeom::unfused)
    unsigned char *const dst_u = reinterpret_cast<unsigned char *>(a.dst);
    auto store_unfused = [&](auto fast_tag, int ob, const v16i(&acc)[PXW]) {
      constexpr bool FAST = decltype(fast_tag)::value;
      // (a u8 result saturates at 0 whatever the ReLU flag says: the pack helpers take that as relu = true)
      const bool relu0 = a.relu0 != 0 || DST == DFX_U8;
      const bool fma0 = DST == DFX_U8 && g.m0 != 0;  // "fma" route (accumulators started from the comp slot, bias slot = -2^23 * scale)
#pragma unroll
      for (int p = 0; p < PXW; ++p) {
        const int slot0 = 32 * (wp * PXW + p);
        v4i qv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = ob * 32 + 8 * q + 4 * h;
          const v4f bs4 = *reinterpret_cast<const v4f *>(bias0 + ch);
          const v4f sc4 = *reinterpret_cast<const v4f *>(scale0 + ch);
          v4i cp4 = {0, 0, 0, 0};
          if (!FAST) cp4 = *reinterpret_cast<const v4i *>(comp0 + ch);
          int a4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) a4[i] = acc[p][4 * q + i];
          qv[q] = pw_quarter<DST, FAST>(a4, cp4, bs4, sc4, relu0, a.rm0, fma0);
        }
        if constexpr (ESZ == 1) {  // natural channel order in `mid`; the rows leave after the barrier
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<int *>(mid + (slot0 + l31) * g.mid_stride + ob * 32 + 8 * q + 4 * h) = qv[q][0];
        } else {
          // 4-byte outputs: this block's 32 px x 128 B through a wave-private piece of the (otherwise unused) mid
          // area, then 16 bytes per lane: eight lanes write one pixel's 128 contiguous bytes.  (Stored straight from
          // the accumulators -- 32-byte pieces, four instructions per line -- vgg3 s32 took 197 us against 174 on
          // conv_stream.cuh.)
          unsigned char *stg4 = mid + wave * (32 * 144);
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<v4i *>(stg4 + l31 * 144 + 32 * q + 16 * h) = qv[q];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ck = lane + 64 * k, row = ck >> 3, c16 = ck & 7;
            const unsigned off = pxoff[slot0 + row];
            const v4i val = *reinterpret_cast<const v4i *>(stg4 + row * 144 + 16 * c16);
            if (off != 0xffffffffu && ob * 32 + 4 * c16 < a.oc)
              DFX_STORE16(reinterpret_cast<v4i *>(dst_u + DK_CHK(12, (long long)(off + (unsigned)(ob * 128 + 16 * c16)), 16, g.dst_bytes)), val);
          }
        }
      }
    };
    // T9 (3x3 kernels, DK_RD = 9: one round = the nine taps of one 32-channel input block): the round's loop is
    // free of scalar bookkeeping.  The generic loop tracks (tap, kernel column, input block) and the ring's
    // (output block, k-block) with ~30 scalar instructions per k-block; a SIMD issues about one scalar instruction
    // per 4 cycles over its waves, so two waves spent ~260 cycles of scalar issue per 256 (npb4) or 64 (npb1!)
    // cycles of MFMA.  Here the tap of step i is i: the three kernel rows keep their own fragment addresses
    // (arow[row][pixel block], advanced to the next input block right after the row's last prefetch of the round),
    // the kernel column is the immediate offset of the ds_read, and the ring refills are rbase + i KB.
    auto conv0_stage = [&](auto even_tag, auto t9_tag) {
      constexpr bool EVEN = decltype(even_tag)::value, T9 = decltype(t9_tag)::value;
      static_assert(!T9 || (EVEN && DK_RD == 9), "the 3x3 fast path walks whole rounds of nine taps");
      int arow[3][PXW];
      int icb_r = 0, f_ob = wo, f_kb0 = 0;  // the round's input block; the next REFILL round's output block / first k-block
      int l_tap = 0, l_tkw = 0, l_toff = 0, l_icb = 0;  // position of the NEXT k-block whose pixel fragments get loaded (runs two ahead)
      v4i fb[3][PXW];
#define DK_LOAD_FB(SET)                                                                 \
  do {                                                                                  \
    const int koff_ = __builtin_amdgcn_readfirstlane((l_icb >> 1) * g.plane_bytes + (l_icb & 1) * 32 + l_toff); \
    _Pragma("unroll") for (int p = 0; p < PXW; ++p)                                     \
      fb[SET][p] = *reinterpret_cast<const v4i *>(tile0 + fbyte[p] + koff_);            \
    l_toff += DK_POS;                                                                   \
    if (++l_tkw == a.kw) { l_tkw = 0; l_toff += row_skip; }                             \
    if (++l_tap == ntap) { l_tap = 0; l_tkw = 0; l_toff = 0; if (++l_icb == g.icb) l_icb = 0; } \
  } while (0)
      if (EVEN) {
#pragma unroll
        for (int i = 0; i < DK_RD; ++i) wr[i] = wload(r_next());
        if constexpr (T9) {
          f_ob = r_ob; f_kb0 = r_kb;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int p = 0; p < PXW; ++p) arow[r][p] = fbyte[p] + r * g.row_pitch;
#pragma unroll
          for (int p = 0; p < PXW; ++p) {
            fb[0][p] = *reinterpret_cast<const v4i *>(tile0 + arow[0][p]);
            fb[1][p] = *reinterpret_cast<const v4i *>(tile0 + arow[0][p] + DK_POS);
          }
        } else {
          DK_LOAD_FB(0);
          DK_LOAD_FB(1);
        }
      }
      for (int ob = wo; ob < g.ocb; ob += WO) {
        DFX_STAMP(t1b);
        if (!EVEN) {
          r_ob = ob; r_kb = 0;
#pragma unroll
          for (int i = 0; i < DK_RD; ++i) wr[i] = wload(r_next());
          l_tap = l_tkw = l_toff = l_icb = 0;
          DK_LOAD_FB(0);
          DK_LOAD_FB(1);
        }
        v16i acc[PXW];
        // START_C: the specialised kernels' 3x3 path hands the start values to the first k-block's MFMAs as their C
        // operand instead of copying them into every accumulator first (16 v_mov per pixel block and output block)
        constexpr bool START_C = T9 && QM != 0;
        v16i st = zero16;
        if (QM != 0 || g.m0) {  // "fma": start from bits(2^23) + comp + bias of this lane's 16 channels (comp slot of the constants)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4i iv = *reinterpret_cast<const v4i *>(comp0 + ob * 32 + 8 * q + 4 * h);
            st[4 * q + 0] = iv[0]; st[4 * q + 1] = iv[1]; st[4 * q + 2] = iv[2]; st[4 * q + 3] = iv[3];
          }
        }
        if (!START_C) {
#pragma unroll
          for (int p = 0; p < PXW; ++p) acc[p] = st;
        }
        DKF();
        int kb0 = 0;
        if constexpr (T9) {
          auto round9 = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const int nxt = icb_r + 1 == g.icb ? 0 : icb_r + 1;
            const int dnext = ((nxt >> 1) - (icb_r >> 1)) * g.plane_bytes + ((nxt & 1) - (icb_r & 1)) * 32;
            const int rbase = (min(f_ob, ob_last) * nkb0 + f_kb0) << 10;
#pragma unroll
            for (int i = 0; i < DK_RD; ++i) {
#pragma unroll
              for (int p = 0; p < NF; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], (FIRST && i == 0) ? st : acc[p]);  // D0[oc][px]
              DKF();
              {
                const int t = (i + 2) % 9, rr = t / 3, dx = t % 3;  // the tap two steps ahead (7, 8: the next round's 0, 1); constants once unrolled
#pragma unroll
                for (int p = 0; p < PXW; ++p)
                  fb[(i + 2) % 3][p] = *reinterpret_cast<const v4i *>(tile0 + arow[rr][p] + dx * DK_POS);
              }
              DKF();
#pragma unroll
              for (int p = NF; p < PXW; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], (FIRST && i == 0) ? st : acc[p]);
              DKF();
              wr[i] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(w0rs, (int)lane16, rbase + i * 1024, 0));
              if (i == 0 || i == 3 || i == 6) {  // row i / 3 has issued its last prefetch of this round
#pragma unroll
                for (int p = 0; p < PXW; ++p) arow[i / 3][p] += dnext;
              }
              DKF();
            }
            icb_r = nxt;
            f_kb0 += DK_RD;
            if (f_kb0 == nkb0) { f_kb0 = 0; f_ob += WO; }
          };
          if constexpr (START_C) {  // (nkb0 is a positive multiple of nine on this path)
            round9(TT{});
            kb0 = DK_RD;
          }
          for (; kb0 < nkb0; kb0 += DK_RD) round9(FF{});
        }
        for (; !T9 && kb0 + DK_RD <= nkb0; kb0 += DK_RD) {
#pragma unroll
          for (int i = 0; i < DK_RD; ++i) {
#pragma unroll
            for (int p = 0; p < NF; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], acc[p]);  // D0[oc][px]
            DKF();
            DK_LOAD_FB((i + 2) % 3);
            DKF();
#pragma unroll
            for (int p = NF; p < PXW; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], acc[p]);
            DKF();
            wr[i] = wload(r_next());  // refill the ring slot (returns ~1k cycles later)
            DKF();
          }
        }
        if (!EVEN) {
#pragma unroll
          for (int i = 0; i < DK_RD; ++i) {
            if (kb0 + i < nkb0) {
#pragma unroll
              for (int p = 0; p < PXW; ++p) acc[p] = mfma_i8(wr[i], fb[i % 3][p], acc[p]);
              if (kb0 + i + 2 < nkb0) DK_LOAD_FB((i + 2) % 3);
            }
          }
        }
        DFX_STAMP(t2);
        if constexpr (QM != 0) {
          requant0(std::integral_constant<int, 2>{}, ob, acc);
        } else {
          if (g.unfused) {
            if (fast) store_unfused(TT{}, ob, acc); else store_unfused(FF{}, ob, acc);
          } else if (g.m0) requant0(std::integral_constant<int, 2>{}, ob, acc);
          else if (fast) requant0(std::integral_constant<int, 1>{}, ob, acc);
          else requant0(std::integral_constant<int, 0>{}, ob, acc);
        }
        DFX_STAMP(t3);
        DFX_ACC(2, t3 - t2);  // requant 0
        DFX_ACC(1, t2 - t1b);  // conv0 K loop
      }
#undef DK_LOAD_FB
    };
    if (even0 && a.kh == 3 && a.kw == 3 && DK_RD == 9) conv0_stage(TT{}, TT{});
    else if (even0) conv0_stage(TT{}, FF{});
    else conv0_stage(FF{}, FF{});

    // ---- conv1: this wave's groups x its PXW1 pixel blocks, PX1 at a time ----
    // Same scheme: the W1 fragments of this wave are one stream of blocks of G fragments (groups wo1, wo1 + WO1,
    // ...; NP1 passes over each; k-blocks 0 .. ocb - 1) through a ring RD1 blocks deep that is primed BEFORE the
    // barrier and runs on under the store epilogues.
    const bool has1 = wo1 < g.n_g1;  // (false for every wave of an unfused op: n_g1 = 0)
    const int g1_last = has1 ? wo1 + (g.n_g1 - 1 - wo1) / WO1 * WO1 : 0;
    int r_g = wo1, r_pp = 0, r_blk = 0;
    auto r1_next = [&]() -> unsigned {
      const unsigned idx = (unsigned)((min(r_g, g1_last) * g.ocb + r_blk) * G);
      const bool wrap = r_blk + 1 == g.ocb;
      const bool wrap2 = wrap && r_pp + 1 == NP1;
      r_blk = wrap ? 0 : r_blk + 1;
      r_pp = wrap2 ? 0 : (wrap ? r_pp + 1 : r_pp);
      r_g = wrap2 ? r_g + WO1 : r_g;
      return idx;
    };
    auto w1load = [&](unsigned idx) -> v4i {
      return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(w1rs, (int)lane16, (int)(idx << 10), 0));
    };
    v4i wr1[RD1][G];
#pragma unroll
    for (int i = 0; i < RD1; ++i) {
      const unsigned idx = r1_next();
#pragma unroll
      for (int cc = 0; cc < G; ++cc) wr1[i][cc] = w1load(idx + cc);
    }
    DFX_STAMP(t4);
    __syncthreads();  // mid is complete; the tile is dead (the store staging may use it)
    DFX_STAMP(t5);
    DFX_ACC(3, t5 - t4);  // barrier after conv0
    if (g.unfused) {
      if constexpr (ESZ == 1) {  // whole pixel rows of the unit leave `mid`: 16 bytes per lane, a.oc contiguous bytes per pixel
        const int c16n = a.oc >> 4;
        const float r_c16 = 1.0f / __int2float_rn(c16n);
        for (int q = tid; q < 32 * NPB * c16n; q += DK_THREADS) {
          int c16;
          const int slot = divmod(q, c16n, r_c16, c16);
          const unsigned off = pxoff[slot];
          const v4i val = *reinterpret_cast<const v4i *>(mid + slot * g.mid_stride + 16 * c16);
          if (off != 0xffffffffu) DFX_STORE16(reinterpret_cast<v4i *>(dst_u + DK_CHK(11, (long long)(off + 16u * c16), 16, g.dst_bytes)), val);
        }
      }
    }

    if (has1) {
      v4i fa[2][PX1];
#pragma unroll
      for (int p = 0; p < PX1; ++p)
        fa[0][p] = *reinterpret_cast<const v4i *>(mid + (32 * (wp1 * PXW1 + p) + l31) * g.mid_stride + h * 16);
      for (int g1 = wo1; g1 < g.n_g1; g1 += WO1) {
        const int chb = 32 * G * g1 + G * l31;
#pragma unroll
        for (int pp = 0; pp < NP1; ++pp) {
          DFX_STAMP(t6);
          const int pb0 = wp1 * PXW1 + pp * PX1;                        // first pixel block of this pass
          const int pbn = wp1 * PXW1 + ((pp + 1) % NP1) * PX1;          // ... of the next one
          const unsigned char *mid_r[PX1], *mid_n[PX1];
#pragma unroll
          for (int p = 0; p < PX1; ++p) {
            mid_r[p] = mid + (32 * (pb0 + p) + l31) * g.mid_stride + h * 16;
            mid_n[p] = mid + (32 * (pbn + p) + l31) * g.mid_stride + h * 16;
          }
          v16i acc1[PX1][G];
          // "magic": the accumulator's bits read as 1/(2 pi) + raw * 2^-26.  The start value is the C operand of the
          // pass's first MFMAs (an inline constant: no register, no v_mov -- 16 per accumulator and pass otherwise)
          constexpr int m1s = QM != 0 ? MAGIC1_BITS : 0;
          const v16i st1 = {m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s, m1s};
          DKF();
          constexpr int NM = PX1 * G, NF1 = NM < 2 ? NM : 2;
          auto kround1 = [&](int b0, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
            for (int i = 0; i < RD1; ++i) {
              const int blk = b0 + i;
#pragma unroll
              for (int m = 0; m < NF1; ++m)
                acc1[m % PX1][m / PX1] = mfma_i8(fa[i & 1][m % PX1], wr1[i][m / PX1], (FIRST && i == 0) ? st1 : acc1[m % PX1][m / PX1]);
              DKF();
              {  // the next k-block's pixel fragments; behind the last one: the next pass's first
                const bool last = blk + 1 == g.ocb;
#pragma unroll
                for (int p = 0; p < PX1; ++p)
                  fa[(i + 1) & 1][p] = *reinterpret_cast<const v4i *>(last ? mid_n[p] : mid_r[p] + (blk + 1) * 32);
              }
              DKF();
#pragma unroll
              for (int m = NF1; m < NM; ++m)
                acc1[m % PX1][m / PX1] = mfma_i8(fa[i & 1][m % PX1], wr1[i][m / PX1], (FIRST && i == 0) ? st1 : acc1[m % PX1][m / PX1]);
              DKF();
              {
                const unsigned idx = r1_next();
#pragma unroll
                for (int cc = 0; cc < G; ++cc) wr1[i][cc] = w1load(idx + cc);
              }
              DKF();
            }
          };
          kround1(0, TT{});  // (ocb is a positive multiple of RD1)
          for (int b0 = RD1; b0 < g.ocb; b0 += RD1) kround1(b0, FF{});
          DFX_STAMP(t7);
          DFX_ACC(4, t7 - t6);  // conv1 K loop
          // ---- requant 1 + store (constants from LDS: a global load here would wait for the ring) ----
          unsigned char *dst_b = reinterpret_cast<unsigned char *>(a.dst);
          const unsigned chbE = (unsigned)chb * ESZ;
          // MODE 3: "fma", 2: "magic" (u8 through the 16-byte store path only; bias / scale slots hold the mode's B / C,
          // conv_mfma.cuh emit_pair), 1: fast, 0: exact
          auto emit = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool FAST = MODE != 0;
            int cp[G];
            float bs[G], sc[G], zf[G];
#pragma unroll
            for (int cc = 0; cc < G; ++cc) {
              cp[cc] = FAST ? 0 : comp1[chb + cc];
              bs[cc] = bias1[chb + cc];
              sc[cc] = scale1[chb + cc];
              zf[cc] = 0.0f;
            }
            if constexpr (MODE >= 2 && !DK_QM_TRANSPOSE) {
              // conversion-free routes: straight from the accumulators, one dword (this lane's 4 channels) per pixel;
              // a wave instruction writes two pixels' 128 contiguous bytes.  No LDS round trip, no predicate: the slot
              // table points the slots behind the unit's last pixel AT that pixel, and those slots hold exactly its
              // values (they read its input positions in conv0), so their stores repeat its bytes.
              if (chb < a.oc1) {
#pragma unroll
                for (int p = 0; p < PX1; ++p) {
                  v4i o4[4];
#pragma unroll
                  for (int eq = 0; eq < 4; ++eq) o4[eq] = *reinterpret_cast<const v4i *>(pxoff + 32 * (pb0 + p) + 8 * eq + 4 * h);
#pragma unroll
                  for (int e = 0; e < 16; ++e) {
                    unsigned pk = 0;
#pragma unroll
                    for (int cc = 0; cc < G; ++cc) {
                      const float x = __int_as_float(acc1[p][cc][e]);
                      pk = __builtin_amdgcn_cvt_pk_u8_f32(MODE == 3 ? __builtin_fmaf(x, sc[cc], bs[cc]) : __fmul_rn(__fadd_rn(x, bs[cc]), sc[cc]), cc, pk);
                    }
                    const unsigned off = (unsigned)o4[e >> 2][e & 3] + chbE;
                    DFX_STORE(reinterpret_cast<unsigned *>(dst_b + DK_CHK(11, (long long)off, 4, g.dst_bytes)), pk);
                  }
                }
              }
            } else if constexpr (ESZ == 1 && G == 4) {
              // 1-byte outputs: transpose 32 px x 128 B through LDS, 16-byte stores (see conv_stream.cuh)
              unsigned char *stg = tile0 + wave * DK_STAGE;
#pragma unroll
              for (int p = 0; p < PX1; ++p) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                  unsigned pk = 0;
                  if constexpr (MODE == 3) {
#pragma unroll
                    for (int cc = 0; cc < G; ++cc)
                      pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(__int_as_float(acc1[p][cc][e]), sc[cc], bs[cc]), cc, pk);
                  } else if constexpr (MODE == 2) {
#pragma unroll
                    for (int cc = 0; cc < G; ++cc)
                      pk = __builtin_amdgcn_cvt_pk_u8_f32(__fmul_rn(__fadd_rn(__int_as_float(acc1[p][cc][e]), bs[cc]), sc[cc]), cc, pk);
                  } else {
                    int v[G];
#pragma unroll
                    for (int cc = 0; cc < G; ++cc) v[cc] = acc1[p][cc][e] + cp[cc];
                    pk = pack_group<DST, G, FAST>(v, zf, bs, sc, relu1, a.rm1);
                  }
                  *reinterpret_cast<unsigned *>(stg + (8 * (e >> 2) + (e & 3) + 4 * h) * 144 + 4 * l31) = pk;
                }
                // all eight LDS reads first (one wait), then the stores: inside the stores' exec-masked regions
                // hipcc issued them one by one, two dependent LDS round trips per store
                unsigned off4[4];
                v4i val4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int c = lane + 64 * k, px = c >> 3, c16 = c & 7;
                  off4[k] = pxoff[32 * (pb0 + p) + px];
                  val4[k] = *reinterpret_cast<const v4i *>(stg + px * 144 + 16 * c16);
                }
                DKF();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const int c16 = lane & 7;
                  if (off4[k] != 0xffffffffu && 128 * g1 + 16 * c16 < a.oc1)
                    DFX_STORE16(reinterpret_cast<v4i *>(dst_b + DK_CHK(11, (long long)(off4[k] + 128 * g1 + 16 * c16), 16, g.dst_bytes)), val4[k]);
                }
                DKF();
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
                        store_group<DST, G, FAST>(dst_b + DK_CHK(12, (long long)(off + chbE), G * ESZ, g.dst_bytes), v, zf, bs, sc,
                                                  relu1, a.rm1);
                      }
                    }
                  }
              }
            }
          };
          if constexpr (QM == 2) {
            emit(std::integral_constant<int, 3>{});
          } else if constexpr (QM == 1) {
            emit(std::integral_constant<int, 2>{});
          } else {
            if (fast) emit(std::integral_constant<int, 1>{});
            else emit(std::integral_constant<int, 0>{});
          }
          DFX_STAMP(t8);
          DFX_ACC(5, t8 - t7);  // requant 1 + stores
        }
      }
    }
    DFX_STAMP(t9);
    DFX_ACC(6, t9 - t0);
#ifdef DFX_STAMPS
    prof_acc[9] = t9 - t_entry;  // kernel entry -> end of this wave's (so far) last unit
#endif
    DFX_ACC(7, 1);
  }
#ifdef DFX_STAMPS
  if (lane == 0) {
    unsigned long long *o = g.prof + ((size_t)blockIdx.x * NW + wave) * 16;
    for (int k = 0; k < 10; ++k) o[k] = prof_acc[k];
  }
#endif
#undef DK_T_ISSUE
#undef DK_T_COMMIT
}

#undef DKF

}  // namespace dfx
