// conv_mfma_roles.cuh -- the fused u8 x s8 conv3x3 + ReLU + requant + conv1x1 (+ReLU) + requant
// block with 1-byte output, as a pipeline of SPECIALISED waves inside one persistent workgroup per CU
// (gfx950 / CDNA4).  Same arithmetic, LDS image, unit / tile protocol and loader as conv_mfma.cuh; what
// changes is who does what.
//
// Replaces (like conv_mfma.cuh):
//   compute_loop / store_output        /root/reference/src/jit_conv_kernel.cc:317-393, :218-305
//   compute1x1_loop / store_1x1output  src/jit_conv_kernel.cc:143-191, :50-141
//   infer_conv0conv1                   src/op_conv.cc:140-260
//
// Why (round 3 measurements, profiles/r03/):
//  * In conv_mfma.cuh every compute wave walks conv0 (36 MFMAs, LDS-fed) -> requant 0 -> conv1 (16 MFMAs) ->
//    requant 1 + stores one after the other.  A wave issues in order, so its matrix work and its vector work never
//    overlap and every phase change exposes an LDS round trip (first fragments, 1x1 weights twice, constants).
//  * What the hardware can overlap, measured with tools/probe/probe_coexec.hip and probe_issue.hip: a SIMD issues
//    ONE vector instruction (VALU or MFMA) per 4 cycles over all of its waves, a scalar one and an LDS one beside
//    it from other waves; plain VALU instructions run under an MFMA in flight (six per 32-cycle MFMA in one wave's
//    stream cost nothing, another wave's stream runs at 6.1 instead of 5.3 cycles per instruction), the PACKED f32
//    forms do not (round 2's v_pk_add / v_pk_mul epilogue, see DFX_PACKED_F32 in conv_mfma.cuh); one wave alone
//    issues at most one instruction per ~5 cycles of whatever kind.
//  So: the fewest possible vector instructions per output value (one v_fma_f32 + one v_cvt_pk_u8_f32), no address
//  or control arithmetic in vector registers, and the phases as ROLES with their own waves, so that a SIMD always
//  holds matrix-heavy waves next to vector-heavy ones:
//   * 6 A waves: conv0 + requant 0.  36 MFMAs per 32-pixel tile back to back, fragments prefetched through a
//     register ring.  The u8 intermediate (32 pixels x OC bytes = 2 KB, in the 1x1 MFMA's A-fragment order) goes
//     to a ring of `mid` slots in LDS.  The reference keeps it in xmm registers (jit_conv_kernel.cc:275-277); here
//     it crosses from one wave's registers to another's through LDS and never reaches HBM either.
//   * 8 B waves: conv1 + requant 1 + stores.  Wave j serves channel group j % NCG (128 output channels) of every
//     tile: its 1x1 weight fragments (from global memory, once) and its lane's requant constants stay RESIDENT IN
//     REGISTERS for the whole launch, so a B wave's LDS traffic per visit is the 2 KB of `mid` and three words.
//   * 2 loader waves: conv_mfma.cuh's (global -> registers -> LDS halo tiles, units from the static split / queue).
//   16 waves x 128 VGPRs.  LDS traffic per tile drops from 72 KB to ~58 KB (no 1x1 weight or start-value re-reads).
// Hand-offs are LDS words written and read by whole waves, all control flow scalar (see conv_mfma.cuh):
//   A: tile claim t (ds_append CTL_NEXT) -> unit record of the slot (ONE ds_read_b128: {dst pixel, th/tw, tiles per
//      row, generation published}) -> conv0 -> count the claim off its input slot (CTL_DONE, lane-0 ds_add) ->
//      requant 0 -> mid slot t % NM (generation t / NM; wait MFREE) -> write mid + {dst pixel, valid pixels} ->
//      MFULL.  EVERY claim that is not a wave's last publishes a slot, if only an empty record (valid pixels 0), so
//      the slots can be walked in claim order and no head counter exists.
//   B: b = next claim of its group (MTAIL[group], synchronous lane-0 ds_add_rtn placed behind the MFMAs it has just
//      issued) -> wait MFULL[b % NM] -> read mid -> MFREE += 64 -> 2 x 4 MFMAs from registers -> per pixel 4 x
//      (v_fma_f32 or v_add_f32 + v_mul_f32, v_cvt_pk_u8_f32) and one global_store_dword with a scalar base: a
//      half-wave writes one whole 128-byte line, no address instruction.
//   End: an A wave that runs out of tiles adds itself to ADONE; a B wave whose slot stays empty after that leaves.
//   Every spin is bounded (RL_SPIN_LIMIT / MFMA_SPIN_LIMIT).
// The mid ring lives in the LDS area the packed 1x1 weights have in conv_mfma.cuh's LDS image.
// Start-up: W0 + constants travel global -> LDS by LDS-DMA (no VGPR / ds_write detour), the first unit of
// each loader stream is staged by the A / B waves in the same memory round trip; the loader fetches the second
// one behind the barrier (see there for what else was tried).
// Unit numbering is XCD-major (see `wg`): vertically neighbouring units share an L2, HBM reads 1.03 x the input
// instead of 1.5 x (PMC FETCH_SIZE 13.3 MB x 2 against 19.2 MB x 2).
//
// Requant: stage 0 takes the host-proven "fma" mode only (geom.mode0 == 3): accumulators start from
// bits(2^23) + comp + bias, so their bits read as the float 2^23 + t for the true sum t = acc + bias >= 0
// and as 2^23 - |t|/2 (the binade below: still negative after the subtraction) for t < 0; one
// v_fma_f32(x, s, -2^23 s) then yields t*s with the reference's single rounding (2^23 s is exact) for
// t >= 0 and some negative number for t < 0, which the stage's ReLU + unsigned saturation turns into 0
// exactly as they do the reference's negative product.  Needs s >= 0.  Stage 1 takes conv_mfma.cuh's "magic" mode
// (geom.mode1 == 2: add + mul) or, where every channel's addend (comp + bias - m / ulp) * scale is exactly
// representable (power-of-two scales and a few others, proven per channel by the host), one v_fma_f32
// (geom.mode1 == 3).  Everything else (exact x86 overflow semantics, round-down, negative scales, 4-byte outputs)
// stays on conv_mfma.cuh.
//
// Supported: what conv_mfma.cuh supports, with 1-byte output and oc1x1 a multiple of 128 up to 512.
#pragma once

#include "conv_mfma.cuh"

namespace dfx {

#ifndef RL_RING
#define RL_RING 5  // conv0 fragment prefetch depth of an A wave (k-steps in flight)
#endif
#ifndef RL_NA
#define RL_NA 6
#endif
#ifndef RL_NB
#define RL_NB 8
#endif
constexpr int RL_A = RL_NA;                          // conv0 waves
constexpr int RL_B = RL_NB;                          // conv1 + store waves
constexpr int RL_C = RL_A + RL_B;                    // waves that stage the weights
constexpr int RL_WAVES = RL_C + MFMA_TEAMS;          // + 2 loaders
constexpr int RL_THREADS = 64 * RL_WAVES;            // 768
constexpr int RL_CTRL_BYTES = 512;                   // control block: the words of conv_mfma.cuh + the mid ring's
constexpr int MAGIC3_BITS = 0x4B000000;              // 2^23: stage-0 accumulator start of the "fma" mode
constexpr int RL_SPIN_LIMIT = 1 << 19;               // bound of the s_sleep(4) waits (~0.1 s): a protocol error ends the launch with
                                                     // wrong output (the parity tests catch it) instead of hanging the GPU

// control words (ints) behind those of conv_mfma.cuh (CTL_*, < 32)
constexpr int RCTL_ADONE = 33;   // 64 x A waves that have left
constexpr int RCTL_MTAIL = 36;   // [4] per channel group: 64 x mid slots claimed by the group's B waves
constexpr int RCTL_MFULL = 40;   // [8] generations published into mid slot s
constexpr int RCTL_MFREE = 48;   // [8] 64 x reads of mid slot s counted off (NCG per generation: one per channel group)
constexpr int RCTL_MINFO = 64;   // [8][4] per mid slot: dst pixel index of the tile's first pixel, valid pixels
// Input-tile slots: the unit record CTL_INFO[slot] = {dst pixel of the unit's first pixel, (th << 16) | tw,
// ceil(2^32 / tiles per row), generations PUBLISHED} -- the loader writes the first three words, then the fourth;
// an A wave takes all four with ONE ds_read_b128 (a lane's 16 bytes are read in one LDS cycle: if the fourth word
// shows the generation it waits for, the three before it belong to that generation).  CTL_FULL is not used here.

// One dword per lane to (uniform base + per-lane byte offset + immediate): the scalar-base form of
// global_store_dword, so that a store costs no address instruction at all.  (Given `base + lane offset` as a
// pointer hipcc builds the 64-bit address with a v_lshl_add_u64 per store; a SIMD issues about one instruction
// per 4 cycles over all of its waves -- SQ_ACTIVE_INST_ANY covers the whole launch -- so every instruction
// removed from the tile loop shortens the kernel.)  Dword stores need no wait states after them (conv_mfma.cuh).
template <int IMM>
__device__ __forceinline__ void store_dword_saddr_nt(const void *sbase, unsigned voff, unsigned data) {
  static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate");
  asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" ::"v"(voff), "v"(data), "s"(sbase), "n"(IMM) : "memory");
}

// Control-word updates by LANE 0 ALONE (the wave is fully active at every call site; exec is restored to all
// ones).  `__hip_atomic_fetch_add` from all 64 lanes is what conv_mfma.cuh uses; hipcc turns it into the same
// single-lane add, but through v_mbcnt / v_cmp / s_and_saveexec / branch / s_bcnt1 / v_mov -- a dozen instructions
// per update, four updates per tile here.  `addr` / `val` are VGPRs holding the word's LDS byte address (the control
// block sits at LDS address 0) and the addend.
__device__ __forceinline__ void lds_add_lane0(int addr, int val) {
  asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(val) : "memory");
}
// Returning form, SYNCHRONOUS: the wait for the result is inside the asm statement.  (An asm-issued ds_add_rtn
// delivers its result asynchronously into a register hipcc believes written already; with the wait in a second
// statement a copy hipcc inserted in between -- for a loop-carried value -- read stale data: found as randomly
// skipped tiles.)  Place it where the wave has to wait anyway, e.g. behind freshly issued MFMAs.
__device__ __forceinline__ int lds_add_rtn_lane0_sync(int addr, int val) {
  int r;
  asm volatile("s_mov_b64 exec, 1\n\tds_add_rtn_u32 %0, %1, %2\n\ts_mov_b64 exec, -1\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(r) : "v"(addr), "v"(val) : "memory");
  return __builtin_amdgcn_readfirstlane(r);
}

// Stamps build only (make stamps; never quote its run time): every wave keeps cycle sums in scalar registers and
// writes them at exit to g.prof[(workgroup * 16 + wave) * 16 + k]; profiles/stamps_roles.py prints them.
#ifdef DFX_STAMPS
#define RL_SUMS unsigned long long rl_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define RL_ADD(k, v) rl_sum[k] += (v)
#define RL_FLUSH(role)                                                                                   \
  do {                                                                                                   \
    unsigned long long rt_;                                                                              \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory");                      \
    const unsigned long long te_ = dfx_stamp();                                                          \
    if (lane == 0) {                                                                                     \
      unsigned long long *o_ = g.prof + ((size_t)blockIdx.x * 16 + wave) * 16;                           \
      for (int k_ = 0; k_ < 12; ++k_) o_[k_] = rl_sum[k_];                                               \
      o_[12] = t_entry; o_[13] = te_; o_[14] = rt_; o_[15] = (role);                                     \
    }                                                                                                    \
  } while (0)
#else
#define RL_SUMS
#define RL_ADD(k, v)
#define RL_FLUSH(role)
#endif

template <int ICB, int OCB, int NCB, int DST>
__global__ __launch_bounds__(RL_THREADS, (RL_WAVES + 3) / 4) void conv_mfma_roles_kernel(ConvArgs a, MfmaGeom g) {
  constexpr int IC = 32 * ICB, OC = 32 * OCB, CP = IC / 16, G = 4, NCG = NCB / G, OC1 = 32 * NCB;
  constexpr int NM = NCB >= 8 ? 8 : 4, LOG_NM = NCB >= 8 ? 3 : 2;  // mid slots (OCB KB each) inside the W1 area
  static_assert(DST == DFX_U8 || DST == DFX_S8, "1-byte outputs");
  static_assert(NCB % G == 0 && NCG <= 4 && NCG <= RL_B && NM <= NCB, "one B wave per group of 128 output channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  DFX_STAMP(t_entry);
  RL_SUMS;

  // LDS: [control block | W0 fragments | W1 fragments -> mid ring | constants | MFMA_NB input tiles]
  int *ctrl = reinterpret_cast<int *>(smem);
  unsigned char *w0s = smem + RL_CTRL_BYTES;                   // [OCB][9][ICB][64 lanes][16 B]
  unsigned char *w1s = w0s + OCB * 9 * ICB * 1024;             // [NCB][OCB][64 lanes][16 B]
  float *cst = reinterpret_cast<float *>(w1s + NCB * OCB * 1024);
  constexpr int cst_bytes = (mfma_cst_floats(OC, OC1) * 4 + 15) & ~15;
  unsigned char *tiles = reinterpret_cast<unsigned char *>(cst) + cst_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int LW = g.tw + 2;
  const int upi = g.uy * g.ux;
  // Workgroup id with XCD-major numbering: hardware deals workgroups round-robin over the 8 XCDs, so the
  // units of consecutive ids -- vertical neighbours that share two halo rows -- land on eight different L2s
  // and every halo row is fetched from HBM twice.  Renumbered (blockIdx % 8 picks the group, speed only),
  // neighbours run on one XCD in the same round and the second reader hits in L2.
  const int wg = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8)
                                      : (int)blockIdx.x;

  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
  // ---- halo-tile / unit helpers: as in conv_mfma.cuh ----
  auto load_granule = [&](const uint8_t *src_n, int y0, int x0, int q) {
    const int lr = (int)__umulhi((unsigned)q, g.row_magic);
    const int c = q - lr * g.row_chunks;
    const int X = c / CP, j = (c % CP) ^ chunk_swizzle<CP>(X);
    const int iy = y0 + lr, ix = x0 + X;
    const bool ok = iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw;
    const int cy = min(max(iy, 0), a.ih - 1), cx = min(max(ix, 0), a.iw - 1);
    const unsigned off = (unsigned)((cy * a.iw + cx) * IC + 16 * j);
    const v4i v = *reinterpret_cast<const v4i *>(src_n + off);
    return (ok ? v : v4i{0, 0, 0, 0}) ^ x80;  // stored form: u8 - 128; padding = 0x80
  };
  auto unit_split = [&](int unit, int &n, int &uyi, int &uxi) {
    n = g.upi_magic ? (int)__umulhi((unsigned)unit, g.upi_magic) : unit;
    const int u = unit - n * upi;
    uyi = g.ux_magic ? (int)__umulhi((unsigned)u, g.ux_magic) : u;
    uxi = u - uyi * g.ux;
  };
  auto unit_origin = [&](int unit, const uint8_t *&src_n, int &y0, int &x0) {
    int n, uyi, uxi;
    unit_split(unit, n, uyi, uxi);
    y0 = uyi * g.th - a.pt;
    x0 = uxi * g.tw - a.pl;
    src_n = a.src + (size_t)n * a.ih * a.iw * IC;
  };
  auto unit_info = [&](int unit, int &pix0, int &thtw, int &tprm) {
    int n, uyi, uxi;
    unit_split(unit, n, uyi, uxi);
    const int y0 = uyi * g.th, x0 = uxi * g.tw;
    const int th = min(g.th, a.oh - y0), tw = min(g.tw, a.ow - x0);
    const int tpr = (tw + 31) >> 5;
    pix0 = (n * a.oh + y0) * a.ow + x0;
    thtw = (th << 16) | tw;
    tprm = tpr > 1 ? (int)(((1ull << 32) + tpr - 1) / tpr) : 0;
  };
  auto stream_id = [&](int tm) { return tm * (int)gridDim.x + wg; };
  const bool coop0 = g.static_rounds >= 1;
  auto ctl_load = [&](int idx) {
    return __builtin_amdgcn_readfirstlane(
        __hip_atomic_load(ctrl + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
  };
  auto ctl_store = [&](int idx, int v) {
    __hip_atomic_store(ctrl + idx, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  if (wave >= RL_C) {
    // =========================== loader wave `team` (conv_mfma.cuh's, unchanged) ===========================
    const int team = wave - RL_C;
#ifndef DFX_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    v4i pf[MFMA_LC];
    static_assert(MFMA_LC % 2 == 0, "rel table packs two entries per register");
    unsigned relp[MFMA_LC / 2];
#pragma unroll
    for (int i = 0; i < MFMA_LC; ++i) {
      const int q = min(lane + 64 * i, g.tile_chunks - 1);
      const int lr = (int)__umulhi((unsigned)q, g.row_magic);
      const int c = q - lr * g.row_chunks;
      const int X = c / CP;
      const unsigned r16 = (unsigned)((lr * a.iw + X) * CP + ((c % CP) ^ chunk_swizzle<CP>(X)));
      if (i % 2 == 0) relp[i / 2] = r16 & 0xffffu;
      else relp[i / 2] |= r16 << 16;
    }
    const int pad_half = (g.th + 2) * CP;
    int padoff[2], padside[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int e = lane + 64 * m, side = e >= pad_half ? 1 : 0, r = e - side * pad_half;
      padside[m] = e < 2 * pad_half ? side : 2;  // 2 = no entry
      padoff[m] = (((r / CP) * LW + (side ? LW - 1 : 0)) * CP + (r % CP)) * 16;
    }
    const bool fast_ok = 2 * pad_half <= 128 && (g.th + 2) * a.iw * CP < 65536;
    auto unit_fast = [&](int unit, const uint8_t *&base_u, int &left, int &right) {
      const uint8_t *src_n;
      int y0, x0;
      unit_origin(unit, src_n, y0, x0);
      base_u = src_n + ((long long)y0 * a.iw + x0) * IC;
      left = x0 < 0;
      right = x0 + LW - 1 >= a.iw;
      return fast_ok && y0 >= 1 && y0 + g.th + 1 <= a.ih - 2 && x0 >= -1 && x0 + LW - 1 <= a.iw;
    };
    int cur_fast = 0, cur_left = 0, cur_right = 0;
#define DFX_PREFETCH(UNIT)                                                              \
  do {                                                                                  \
    const uint8_t *base_u_;                                                             \
    cur_fast = unit_fast((UNIT), base_u_, cur_left, cur_right) ? 1 : 0;                 \
    if (cur_fast) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < MFMA_LC; ++i) {                             \
        unsigned rp_ = relp[i / 2];                                                     \
        asm volatile("" : "+v"(rp_));                                                   \
        pf[i] = *reinterpret_cast<const v4i *>(base_u_ + ((i % 2 ? rp_ >> 16 : rp_ & 0xffffu) << 4)); \
      }                                                                                 \
    } else {                                                                            \
      const uint8_t *src_n_;                                                            \
      int y0_, x0_;                                                                     \
      unit_origin((UNIT), src_n_, y0_, x0_);                                            \
      int lq_ = lane;                                                                   \
      asm volatile("" : "+v"(lq_));                                                     \
      _Pragma("unroll") for (int i = 0; i < MFMA_LC; ++i)                               \
          pf[i] = load_granule(src_n_, y0_, x0_, min(lq_ + 64 * i, g.tile_chunks - 1)) ^ x80; \
    }                                                                                   \
  } while (0)
    const int T = (int)gridDim.x * MFMA_TEAMS, tg = stream_id(team);
    const bool use_queue = g.static_rounds * T < g.total_units;
    auto unit_at = [&](int j) {
      int v = j * T + tg;
      if (j >= g.static_rounds) {
        v = 0x7fffffff;
        if (use_queue && lane == 0) v = g.static_rounds * T + atomicAdd(g.queue, 1);
      }
      return v;
    };
    auto write_tile = [&](unsigned char *ins) {
      {
        unsigned char *dst = ins + lane * 16;
        const int dump = g.tile_stride - 1024;
#pragma unroll
        for (int i = 0; i < MFMA_LC; ++i)
          *reinterpret_cast<v4i *>(dst + (64 * i < g.tile_chunks ? 1024 * i : dump)) = pf[i] ^ x80;
      }
      if (cur_fast) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if ((padside[m] == 0 && cur_left) || (padside[m] == 1 && cur_right))
            *reinterpret_cast<v4i *>(ins + padoff[m]) = x80;
      }
    };
    auto write_rest = [&](unsigned char *ins, int unit) {
      // (one granule per trip.  Eight loads in flight per trip -- 32 more live VGPRs in this wave -- cost the u8
      // headline 4.5 us, the s32 one 3.5 us and VGG f32 40 %, although only the last takes this path at all:
      // profiles/r03/ab_loader_rest_batched.txt)
      if (g.tile_chunks > 64 * MFMA_LC) {
        const uint8_t *src_n; int y0, x0;
        unit_origin(unit, src_n, y0, x0);
        for (int q = 64 * MFMA_LC + lane; q < g.tile_chunks; q += 64)
          *reinterpret_cast<v4i *>(ins + 16 * q) = load_granule(src_n, y0, x0, q);
      }
    };
    const int j0 = coop0 ? 1 : 0;
    // The stream's second unit is fetched here, behind the barrier, like every later one.  History (stamps, cycles
    // after entry, headline block): staged before the barrier -- by the loader (round 2) or by the compute waves
    // next to the first units (most of round 3) -- all of the chip's CUs pull 4 units each from HBM at once: the
    // first units arrive after 6.2 k cycles and the workgroup passes the barrier after 9.4 k; fetched behind the
    // barrier, the first units are there after 3.9 k, the barrier is passed after 5.7 k, and the second units are
    // published ~1.3 k cycles after the A waves ran out of first-unit tiles: 28.6 -> 27.9 us
    // (profiles/r03/ab_roles_second_unit.txt).  Issuing these loads BEFORE the barrier and passing it with them in
    // flight (a bare s_barrier) -- timed by a probe load of the first unit's last bytes, or behind a warm-up pass of
    // the prefetch code over the first unit -- made the loader the last wave at the barrier (6.8 k, 8.1 k): its first
    // pass through this code is slow wherever it runs (same file: no gain over staging before the barrier).
    RL_ADD(8, dfx_stamp() - t_entry);  // entry -> reaches the barrier
    if (coop0) __syncthreads();
    RL_ADD(9, dfx_stamp() - t_entry);  // entry -> past the barrier
    const bool lazy = use_queue && g.lazy_queue;
    int cur = lazy ? 0 : __builtin_amdgcn_readfirstlane(unit_at(j0));
    int nxt_v = lazy ? 0 : unit_at(j0 + 1);
    int jn = j0 + 2;
    if (!lazy && cur < g.total_units) DFX_PREFETCH(cur);
    if (!coop0) __syncthreads();  // the only workgroup barrier: weights + control block are in LDS

    RL_ADD(4, dfx_stamp() - t_entry);  // entry -> unit loop (own start-up + the barrier)
    for (int j = j0;; ++j) {
      DFX_STAMP(la);
      const int s = 2 * (j & 1) + team, gen = j >> 1;
      unsigned char *ins = tiles + (size_t)s * g.tile_stride;
      if (lazy) {
        for (int spin = 0; spin < MFMA_SPIN_LIMIT && ctl_load(CTL_DONE + s) < 64 * g.ntu * gen; ++spin)
          __builtin_amdgcn_s_sleep(8);
        cur = __builtin_amdgcn_readfirstlane(unit_at(j));
        if (cur < g.total_units) DFX_PREFETCH(cur);
      }
      const bool valid = cur < g.total_units;
      if (!valid) {
        if (!(coop0 && j == 1 && tg >= g.total_units)) ctl_store(CTL_END + team, 2 * j + team);
        break;
      }
      for (int spin = 0; spin < MFMA_SPIN_LIMIT && ctl_load(CTL_DONE + s) < 64 * g.ntu * gen; ++spin)
        __builtin_amdgcn_s_sleep(2);
      DFX_STAMP(lb);
      write_tile(ins);
      write_rest(ins, cur);
      {
        int i0, i1, i2;
        unit_info(cur, i0, i1, i2);
        ctrl[CTL_INFO + 4 * s + 0] = i0;
        ctrl[CTL_INFO + 4 * s + 1] = i1;
        ctrl[CTL_INFO + 4 * s + 2] = i2;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      ctl_store(CTL_INFO + 4 * s + 3, gen + 1);
      DFX_STAMP(lc);
      if (!lazy) {
        cur = __builtin_amdgcn_readfirstlane(nxt_v);
        nxt_v = unit_at(jn++);
        if (cur < g.total_units) DFX_PREFETCH(cur);
      }
      DFX_STAMP(ld);
      RL_ADD(0, lb - la); RL_ADD(1, lc - lb); RL_ADD(2, ld - lc); RL_ADD(3, 1);  // slot wait, write + publish, draw + prefetch issue, units
    }
#undef DFX_PREFETCH
    RL_FLUSH(2);
    const int pending = __builtin_amdgcn_readfirstlane(nxt_v);
    if (use_queue && lane == 0 && pending >= 0) {
      const int fin = atomicAdd(g.queue + 1, 1);
      if (fin == (int)gridDim.x * MFMA_TEAMS - 1) {
        atomicExch(g.queue, 0);
        atomicExch(g.queue + 1, 0);
      }
    }
    return;
  }

  // =========================== A and B waves: stage weights, constants, first tiles ===========================
  {
    constexpr int CWT = RL_C / 2;        // waves that stage one first tile
    constexpr int TT = CWT * 64;
    const int steam = wave / CWT, ctid = wave * 64 + lane, tctid = (wave % CWT) * 64 + lane;
    const v4i *s = reinterpret_cast<const v4i *>(a.wei);
    v4i *d = reinterpret_cast<v4i *>(w0s);
    const int total = OCB * 9 * ICB * 64 + NCB * OCB * 64 + (mfma_cst_floats(OC, OC1) * 4 + 15) / 16;
    unsigned char *slot = tiles + (size_t)steam * g.tile_stride;
    const int unit0 = stream_id(steam);
    const bool tile0 = coop0 && unit0 < g.total_units;
    const uint8_t *src_n = a.src;
    int y0 = 0, x0 = 0;
    if (tile0) unit_origin(unit0, src_n, y0, x0);
    // ONE memory round trip, and no VGPR / ds_write detour for the weights: the packed weights + constants go
    // global -> LDS by LDS-DMA (global_load_lds_dwordx4: a wave moves 1 KB per instruction, lane i's 16 bytes land
    // at the wave-uniform LDS base + 16 i; exec-masked lanes move nothing), issued before this thread's share of
    // the first tile is loaded (that one needs the xor 0x80, so it travels through registers).  Through ds_write
    // the 56 KB cost every staging wave ~1.5 k cycles of LDS store issue (stamps: loads arrived 3.5 k cycles
    // after entry, LDS writes done 5.0 k).
    {
      typedef __attribute__((address_space(3))) void lds_void;
      typedef __attribute__((address_space(1))) const void global_void;
      // (the image is [W0 | W1 | constants] in global memory and in LDS; the W1 part of LDS is the mid ring, the
      // B waves take their 1x1 fragments from global memory themselves)
      constexpr int w1_first = OCB * 9 * ICB * 64, w1_end = w1_first + NCB * OCB * 64;  // 16-byte chunks
      const int nblk = (total + 63) >> 6;  // 1 KB blocks
      for (int j = wave; j < nblk; j += RL_C) {
        const int q = 64 * j + lane;
        if (q < total && (q < w1_first || q >= w1_end))
          __builtin_amdgcn_global_load_lds((global_void *)(s + q), (lds_void *)(d + 64 * j), 16, 0, 0);
      }
      // The first unit of stream `steam`: all its global loads are issued before the first LDS write.  (The
      // stream's SECOND unit is the loader's, see there.  Earlier versions staged it here as well, in the same memory
      // round trip: the workgroup then waits at the barrier for twice the bytes while the whole chip is pulling
      // its first units from HBM -- barrier passed 9.4 k cycles after entry instead of 5.7 k.  Waiting only for W0 +
      // the first units at the barrier and writing the second units behind it did not help either: the loads still
      // compete, profiles/r03/ab_roles_defer_second_units.txt.)
      constexpr int NTL = 4;  // 16-byte chunks per thread held in registers at once
      v4i tv[NTL];
      if (tile0) {
#pragma unroll
        for (int i = 0; i < NTL; ++i) tv[i] = load_granule(src_n, y0, x0, min(tctid + i * TT, g.tile_chunks - 1));
#pragma unroll
        for (int i = 0; i < NTL; ++i)
          if (tctid + i * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (tctid + i * TT)) = tv[i];
        for (int q = tctid + NTL * TT; q < g.tile_chunks; q += TT)
          *reinterpret_cast<v4i *>(slot + 16 * q) = load_granule(src_n, y0, x0, q);
      }
#ifdef DFX_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RL_ADD(7, dfx_stamp() - t_entry);  // entry -> staged global loads have arrived
#endif
    }
    if (ctid < RL_CTRL_BYTES / 4) {  // control block (see conv_mfma.cuh); the mid ring's words start at zero
      int v = 0;
      if (ctid == CTL_END || ctid == CTL_END + 1) {
        v = 0x7fffffff;
        if (coop0 && stream_id(ctid - CTL_END) >= g.total_units) v = ctid - CTL_END;
      }
      auto first_unit = [&](int sl) {
        const int u = (sl >> 1) * (int)gridDim.x * MFMA_TEAMS + stream_id(sl & 1);
        return sl < 2 && coop0 && u < g.total_units ? u : -1;
      };
      if (ctid >= CTL_INFO && ctid < CTL_INFO + 4 * MFMA_NB) {  // slots staged before the barrier start out published
        const int u0 = first_unit((ctid - CTL_INFO) >> 2);
        if (u0 >= 0) {
          int i0, i1, i2;
          unit_info(u0, i0, i1, i2);
          const int f = (ctid - CTL_INFO) & 3;
          v = f == 0 ? i0 : f == 1 ? i1 : f == 2 ? i2 : 1;
        }
      }
      ctrl[ctid] = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA transfers have landed
  RL_ADD(8, dfx_stamp() - t_entry);  // entry -> own staging done (LDS written)
  __syncthreads();
  RL_ADD(9, dfx_stamp() - t_entry);  // entry -> past the barrier

  typedef __attribute__((address_space(3))) int lds_int;
  const bool relu1 = a.relu1 || DST == DFX_U8;

  if (wave >= RL_A) {
    // =========================== B wave: conv1 + requant 1 + stores ===========================
    // A B wave serves ONE group of G = 4 column blocks (128 output channels) of every tile, wave j group
    // j % NCG: its G x OCB weight fragments and its lane's requant constants (2 floats per channel) are loaded
    // once and stay in registers; a mid slot is read by one wave of every group and free when all NCG have counted
    // it off.  (Whole tiles per wave -- all NCB x OCB fragments resident -- would save the per-visit control
    // instructions of NCG - 1 waves, but needs > 168 VGPRs at the headline shape: hipcc spilled 35.)
    const int cg = (wave - RL_A) % NCG;
    const float *pb1 = cst + 3 * OC + OC1, *pc1 = cst + 3 * OC + 3 * OC1;  // {k, k} pairs (conv_mfma.cuh): [2 * channel]
    constexpr unsigned row_bytes = OC1;  // dst bytes per pixel
    const int l31 = lane & 31, h4 = 4 * (lane >> 5);
    const int chb = 32 * G * cg + G * l31;  // this lane's first channel
    v4i w1r[G][OCB];  // straight from global memory (L2): the 1x1 weights never pass through LDS
#pragma unroll
    for (int cc = 0; cc < G; ++cc)
#pragma unroll
      for (int r = 0; r < OCB; ++r)
        w1r[cc][r] = *reinterpret_cast<const v4i *>(a.wei1 + ((cg * G + cc) * OCB + r) * 1024 + lane * 16);
    float fbk[G], fck[G];
#pragma unroll
    for (int cc = 0; cc < G; ++cc) {
      fbk[cc] = pb1[2 * (chb + cc)];
      fck[cc] = pc1[2 * (chb + cc)];
    }
    const unsigned lane_off = (unsigned)h4 * row_bytes + (unsigned)chb;
    int v64 = 64;  // (one VGPR for the whole loop)
    asm volatile("" : "+v"(v64));
    // Mid slot b holds the CU's b-th tile claim (A waves publish every claim, empty ones too); the waves of a
    // group claim them from the group's counter.
    int b = lds_add_rtn_lane0_sync(4 * (RCTL_MTAIL + cg), v64) >> 6;
    RL_ADD(4, dfx_stamp() - t_entry);  // entry -> tile loop
    for (;;) {
      DFX_STAMP(b0);
      const int bs = b & (NM - 1), bgen = b >> LOG_NM;
      bool have = ctl_load(RCTL_MFULL + bs) >= bgen + 1;
      for (int spin = 0; !have && spin < RL_SPIN_LIMIT; ++spin) {
        // every A wave has left, and each published all of its claims first: nothing more will come
        if (ctl_load(RCTL_ADONE) >= 64 * RL_A) {
          have = ctl_load(RCTL_MFULL + bs) >= bgen + 1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
        have = ctl_load(RCTL_MFULL + bs) >= bgen + 1;
      }
      if (!have) break;
      DFX_STAMP(b1);
      const unsigned char *mslot = w1s + bs * (OCB * 1024);
      v4i mid[OCB];
#pragma unroll
      for (int r = 0; r < OCB; ++r) mid[r] = *reinterpret_cast<const v4i *>(mslot + r * 1024 + lane * 16);
      // (through an LDS-address-space pointer: a volatile vector load through a generic pointer becomes a FLAT load
      // plus s_waitcnt vmcnt(0), i.e. a wait for every store in flight)
      typedef int v2i __attribute__((ext_vector_type(2)));
      typedef __attribute__((address_space(3))) const volatile v2i lds_v2i;
      const v2i minfo = *(lds_v2i *)(ctrl + RCTL_MINFO + 4 * bs);
      // this wave is done with the slot once its reads have returned: LDS serves a wave's operations in order, so
      // the add (and the next claim behind it) cannot overtake them
      lds_add_lane0(4 * (RCTL_MFREE + bs), v64);
      const int obase_i = __builtin_amdgcn_readfirstlane(minfo.x);
      const int nvalid = __builtin_amdgcn_readfirstlane(minfo.y);
      if (nvalid == 0) {  // an empty claim (beyond its unit's tiles, or a unit that does not exist)
        b = lds_add_rtn_lane0_sync(4 * (RCTL_MTAIL + cg), v64) >> 6;
        continue;
      }
      unsigned char *tile_dst = reinterpret_cast<unsigned char *>(a.dst) + (size_t)obase_i * row_bytes;
      const int mode1 = g.mode1;
      {
        v16i acc1[G];
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int cc = 0; cc < G; ++cc)
            acc1[cc] = r == 0 ? mfma_i8_from_magic(mid[0], w1r[cc][0]) : mfma_i8(mid[r], w1r[cc][r], acc1[cc]);

        if (OCB == 1) asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");  // (asm MFMA results: see conv_mfma.cuh)
        // the next claim, behind the MFMAs just issued: its LDS round trip passes while they execute
        b = lds_add_rtn_lane0_sync(4 * (RCTL_MTAIL + cg), v64) >> 6;
        if (DST == DFX_U8 || relu1) {
          // ---- u8: per value one v_fma_f32 (mode 3) or v_add_f32 + v_mul_f32 (mode 2) and one v_cvt_pk_u8_f32
          //      (RNE + [0, 255] saturation = ReLU + vcvtps2dq + vpmovusdb on the values the host admits to these
          //      modes); per pixel one store with a scalar base, no address arithmetic.  Partial tiles (CHECK):
          //      the stores of pixels beyond the tile's end are predicated off.
          //      s8 WITH ReLU takes the same route with one v_med3_f32 more per value: max(0, f) -> vcvtps2dq ->
          //      vpmovsdb (jit_conv_kernel.cc:352-384) lands in [0, 127], which is RNE + u8 saturation of
          //      clamp(f, 0, 127) -- 127.0 converts to 127, anything in (126.5, 127) rounds to it as well ----
          auto fast = [&](auto mode_tag, auto check_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr bool CHECK = decltype(check_tag)::value;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int pl = 8 * (e >> 2) + (e & 3);  // + 4h (in lane_off): pixel of accumulator register e
              unsigned pk = 0;
#pragma unroll
              for (int cc = 0; cc < G; ++cc) {
                const float x = __int_as_float(acc1[cc][e]);
                float f = MODE == 3 ? __builtin_fmaf(x, fck[cc], fbk[cc]) : __fmul_rn(__fadd_rn(x, fbk[cc]), fck[cc]);
                if (DST == DFX_S8) f = __builtin_amdgcn_fmed3f(f, 0.0f, 127.0f);
                pk = __builtin_amdgcn_cvt_pk_u8_f32(f, cc, pk);
              }
              const unsigned off = (unsigned)pl * row_bytes;
              const unsigned char *sb = tile_dst + (off & ~4095u);
              if (!CHECK || pl + h4 < nvalid) {
                switch (off & 4095u) {  // (compile-time after unrolling: pl and row_bytes are constants)
#define RL_CASE(V) case V: store_dword_saddr_nt<V>(sb, lane_off, pk); break;
                  RL_CASE(0) RL_CASE(128) RL_CASE(256) RL_CASE(384) RL_CASE(512) RL_CASE(640) RL_CASE(768) RL_CASE(896)
                  RL_CASE(1024) RL_CASE(1152) RL_CASE(1280) RL_CASE(1408) RL_CASE(1536) RL_CASE(1664) RL_CASE(1792) RL_CASE(1920)
                  RL_CASE(2048) RL_CASE(2176) RL_CASE(2304) RL_CASE(2432) RL_CASE(2560) RL_CASE(2688) RL_CASE(2816) RL_CASE(2944)
                  RL_CASE(3072) RL_CASE(3200) RL_CASE(3328) RL_CASE(3456) RL_CASE(3584) RL_CASE(3712) RL_CASE(3840) RL_CASE(3968)
#undef RL_CASE
                }
              }
            }
          };
          using T = std::true_type;
          using F = std::false_type;
          using M2 = std::integral_constant<int, 2>;
          using M3 = std::integral_constant<int, 3>;
          if (nvalid == 32) { if (mode1 == 3) fast(M3{}, F{}); else fast(M2{}, F{}); }
          else { if (mode1 == 3) fast(M3{}, T{}); else fast(M2{}, T{}); }
        } else {
          // s8 output without ReLU: conv_mfma.cuh's pixel-pair emitter (signed saturation; predicated stores
          // beyond a partial tile's end)
          int ia[G];
          v2f fb[G], fc[G];
#pragma unroll
          for (int cc = 0; cc < G; ++cc) {
            ia[cc] = 0;
            fb[cc] = v2f{fbk[cc], fbk[cc]};
            fc[cc] = v2f{fck[cc], fck[cc]};
          }
          const int nv1 = nvalid - 1;
          const unsigned ch_off = (unsigned)chb;
          auto emit = [&](auto mode_tag) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              const int pl = 8 * (e >> 2) + (e & 3);  // + 4h: pixel of register e; e + 1 is the next pixel
              const bool w0 = pl + h4 <= nv1, w1 = pl + 1 + h4 <= nv1;
              unsigned char *p0 = tile_dst + ((unsigned)min(pl + h4, nv1) * row_bytes + ch_off);
              unsigned char *p1 = tile_dst + ((unsigned)min(pl + 1 + h4, nv1) * row_bytes + ch_off);
              emit_pair<DST, G, decltype(mode_tag)::value>(p0, p1, acc1, e, ia, fb, fc, relu1, a.rm1, w0, w1);
            }
          };
          if (mode1 == 3) emit(std::integral_constant<int, 3>{}); else emit(std::integral_constant<int, 2>{});
        }
      }
      DFX_STAMP(b2);
      RL_ADD(0, b1 - b0); RL_ADD(1, b2 - b1); RL_ADD(3, 1);  // wait for a mid slot, conv1 + requant 1 + store issue, tiles
    }
    RL_FLUSH(1);
    return;
  }

  // =========================== A wave: conv0 + requant 0 -> mid ring ===========================
  const int *ia0 = reinterpret_cast<const int *>(cst);
  const float *fb0 = cst + OC, *fc0 = cst + 2 * OC;  // per conv0 channel: -2^23 * scale, scale
  const int lds_row = LW * IC;                        // bytes per halo-tile row in LDS
  // accumulator start values bits(2^23) + comp + bias of this lane's 16 channels per block: resident, the C
  // operand of each chain's first MFMA
  constexpr bool START_RESIDENT = RL_WAVES <= 12;  // (16 waves: 128 VGPRs per wave, the start values come from LDS per tile)
  v16i start[OCB];
  if (START_RESIDENT) {
    const int h4 = 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < OCB; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4i iv = *reinterpret_cast<const v4i *>(ia0 + 32 * r + 8 * q + h4);
        start[r][4 * q + 0] = iv[0]; start[r][4 * q + 1] = iv[1];
        start[r][4 * q + 2] = iv[2]; start[r][4 * q + 3] = iv[3];
      }
  }
  auto draw = [&]() { return __builtin_amdgcn_ds_append((lds_int *)(ctrl + CTL_NEXT)); };
  // mid-slot claims: an ordinary returning add by all 64 lanes (+64 per claim, lane 0's result = the old value).
  // ds_append takes its address from M0 and is only relied upon for the word at LDS address 0 (CTL_NEXT).
  int v64 = 64;  // (one VGPR for the whole loop)
  asm volatile("" : "+v"(v64));
  struct Look { int end; v4i info; };  // info[3] = generations published into the slot (see RCTL_* above)
  auto look = [&](int sl, int par) {
    Look l;
    typedef __attribute__((address_space(3))) const volatile v4i lds_v4i;  // (LDS pointer: never a flat load)
    l.info = *(lds_v4i *)(ctrl + CTL_INFO + 4 * sl);
    l.end = __hip_atomic_load(ctrl + CTL_END + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return l;
  };
  auto split = [&](int t, int &k, int &ti) {
    k = g.ntu == 1 ? t : (int)__umulhi((unsigned)t, g.ntu_magic);
    ti = t - k * g.ntu;
  };
  int c_ahead = 0;
  int t = __builtin_amdgcn_readfirstlane(draw()) >> 6, k, ti;
  split(t, k, ti);
  Look lk = look(k & (MFMA_NB - 1), k & 1);
  RL_ADD(4, dfx_stamp() - t_entry);  // entry -> tile loop
  for (;;) {
    DFX_STAMP(a0);
    if (t > g.claim_limit) break;  // cannot happen: keeps a logic error from hanging the GPU
    const int s = k & (MFMA_NB - 1), gen = k >> 2, p = k & 1;
    bool have = false;
    v4i info = lk.info;
    for (int spin = 0; spin < RL_SPIN_LIMIT; ++spin) {
      if (__builtin_amdgcn_readfirstlane(lk.info[3]) >= gen + 1) { have = true; info = lk.info; break; }
      if (__builtin_amdgcn_readfirstlane(lk.end) <= k) break;
      __builtin_amdgcn_s_sleep(4);
      lk = look(s, p);
    }
    const int k_cur = k, ti_cur = ti;
    auto next_claim = [&]() {
      t = __builtin_amdgcn_readfirstlane(c_ahead) >> 6;
      split(t, k, ti);
      lk = look(k & (MFMA_NB - 1), k & 1);
    };
    const int t_cur = t;
    // Mid slot of claim t: t % NM, generation t / NM -- every claim that is not the wave's last publishes one, if
    // only an empty record (valid pixels = 0), so that the B waves can walk the slots in claim order.
    auto publish_empty = [&]() {
      const int ms = t_cur & (NM - 1), mgen = t_cur >> LOG_NM;
      for (int spin = 0; spin < RL_SPIN_LIMIT && ctl_load(RCTL_MFREE + ms) < 64 * NCG * mgen; ++spin) __builtin_amdgcn_s_sleep(4);
      typedef int v2i __attribute__((ext_vector_type(2)));
      *reinterpret_cast<v2i *>(ctrl + RCTL_MINFO + 4 * ms) = v2i{0, 0};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      ctl_store(RCTL_MFULL + ms, mgen + 1);
    };
    if (!have) {  // stream p has no k-th unit; done when the other stream has none for k + 1 either
      if (ctl_load(CTL_END + (p ^ 1)) <= k_cur + 1) break;
      publish_empty();
      c_ahead = draw();
      next_claim();
      continue;
    }
    const unsigned char *ins = tiles + (size_t)s * g.tile_stride;
    const int pix0 = __builtin_amdgcn_readfirstlane(info[0]);
    const int thtw = __builtin_amdgcn_readfirstlane(info[1]);
    const int th = thtw >> 16, tw = thtw & 0xffff;
    const int npx = th * tw;
    const int tiles_per_row = (tw + 31) >> 5;
    const int ntiles = g.linear ? (npx + 31) >> 5 : th * tiles_per_row;
    DFX_STAMP(a1);
    RL_ADD(0, a1 - a0);  // wait for the tile's unit
    c_ahead = draw();  // the next tile's claim travels during conv0
    if (ti_cur < ntiles) {
      const int l31 = lane & 31, h = lane >> 5;
      int ty, tx, nvalid, obase;
      if (g.linear) {
        nvalid = min(32, npx - 32 * ti_cur);
        const int pc = 32 * ti_cur + min(l31, nvalid - 1);
        ty = tw == 1 ? pc : (int)__umulhi((unsigned)pc, g.tw_magic);
        tx = pc - ty * tw;
        obase = pix0 + 32 * ti_cur;
      } else {
        const int tprm = __builtin_amdgcn_readfirstlane(info[2]);
        const int tr = tprm ? (int)__umulhi((unsigned)ti_cur, (unsigned)tprm) : ti_cur, tc = ti_cur - tr * tiles_per_row;
        nvalid = min(32, tw - 32 * tc);
        ty = tr;
        tx = 32 * tc + min(l31, nvalid - 1);
        obase = pix0 + tr * a.ow + 32 * tc;
      }
      nvalid = __builtin_amdgcn_readfirstlane(nvalid);
      obase = __builtin_amdgcn_readfirstlane(obase);
      int lane16 = lane * 16;
      asm volatile("" : "+v"(lane16));  // (keeps LICM from hoisting every weight-fragment address)
      // per-lane B-fragment addresses for tap row 0: one per (tap column, ic half)
      int bb[3][ICB];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int X = tx + dx;
        const int pb = (ty * LW + X) * IC, sw = chunk_swizzle<CP>(X);
#pragma unroll
        for (int c = 0; c < ICB; ++c) bb[dx][c] = pb + 16 * ((2 * c + h) ^ sw);
      }
      v16i acc0[OCB];
      if (!START_RESIDENT) {  // start values of this lane's channels from the constant area
        int h4s = 4 * h;
        asm volatile("" : "+v"(h4s));  // (per tile: hoisted, the 32 values would be resident after all)
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4i iv = *reinterpret_cast<const v4i *>(ia0 + 32 * r + 8 * q + h4s);
            acc0[r][4 * q + 0] = iv[0]; acc0[r][4 * q + 1] = iv[1];
            acc0[r][4 * q + 2] = iv[2]; acc0[r][4 * q + 3] = iv[3];
          }
      }
      {
        constexpr int NS = 9 * ICB;  // k-steps
        constexpr int RD = RL_RING;
        v4i fbr[RD], fw[RD][OCB];
        auto fetch = [&](int st, int slot) {  // st, slot are compile-time after unrolling
          const int tap = st / ICB, c = st % ICB;
          // issue order = reverse of the use order (the step's first MFMA takes fw[0] and fbr): one s_waitcnt before
          // the first MFMA then covers the whole step (a wave's LDS reads return in order)
#pragma unroll
          for (int r = OCB - 1; r >= 1; --r)
            fw[slot][r] = *reinterpret_cast<const v4i *>(w0s + ((r * 9 + tap) * ICB + c) * 1024 + lane16);
          fbr[slot] = *reinterpret_cast<const v4i *>(ins + (tap / 3) * lds_row + bb[tap % 3][c]);
          fw[slot][0] = *reinterpret_cast<const v4i *>(w0s + ((0 * 9 + tap) * ICB + c) * 1024 + lane16);
        };
#pragma unroll
        for (int st = 0; st < RD - 1; ++st) fetch(st, st);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < OCB; ++r)
            acc0[r] = mfma_i8(fw[st % RD][r], fbr[st % RD], (st == 0 && START_RESIDENT) ? start[r] : acc0[r]);  // D0[oc][px]
          __builtin_amdgcn_sched_barrier(0);
          if (st + RD - 1 < NS) fetch(st + RD - 1, (st + RD - 1) % RD);
        }
      }
      DFX_STAMP(a2);
      // every fragment of the tile has been consumed (the MFMAs that took the last ones have been issued): count
      // this claim off on its input slot
      lds_add_lane0(4 * (CTL_DONE + s), v64);
      next_claim();
      // ---- requant 0 ("fma" mode) -> A fragments of the 1x1 ----
      v4i mid[OCB];
      const int h4 = 4 * h;
      if (g.s0_uniform) {  // the op's single conv0 scale: a scalar operand, no per-channel reads
        const float su = g.s0_value;
        float cu = -8388608.0f * g.s0_value;
        asm volatile("" : "+v"(cu));  // (one VGPR; v_fma_f32 takes one scalar operand)
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)  // plain v_fma_f32: the packed form does not overlap with MFMAs (conv_mfma.cuh)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(__int_as_float(acc0[r][4 * q + i]), su, cu), i, pk);
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      } else {
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f sc = *reinterpret_cast<const v4f *>(fc0 + 32 * r + 8 * q + h4);
            const v4f cc = *reinterpret_cast<const v4f *>(fb0 + 32 * r + 8 * q + h4);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(__int_as_float(acc0[r][4 * q + i]), sc[i], cc[i]), i, pk);
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      }
      DFX_STAMP(a3);
      // ---- publish into the mid ring ----
      const int ms = t_cur & (NM - 1), mgen = t_cur >> LOG_NM;
      for (int spin = 0; spin < RL_SPIN_LIMIT && ctl_load(RCTL_MFREE + ms) < 64 * NCG * mgen; ++spin) __builtin_amdgcn_s_sleep(4);
      DFX_STAMP(a4);
      unsigned char *mslot = w1s + ms * (OCB * 1024);
#pragma unroll
      for (int r = 0; r < OCB; ++r) *reinterpret_cast<v4i *>(mslot + r * 1024 + lane16) = mid[r];
      typedef int v2i __attribute__((ext_vector_type(2)));
      *reinterpret_cast<v2i *>(ctrl + RCTL_MINFO + 4 * ms) = v2i{obase, nvalid};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // all lanes' writes first
      ctl_store(RCTL_MFULL + ms, mgen + 1);
      DFX_STAMP(a5);
      // address math + conv0, requant 0, wait for a free mid slot, write + publish, tiles
      RL_ADD(1, a2 - a1); RL_ADD(2, a3 - a2); RL_ADD(5, a4 - a3); RL_ADD(6, a5 - a4); RL_ADD(3, 1);
    } else {  // a claim beyond its unit's tiles: nothing to compute
      lds_add_lane0(4 * (CTL_DONE + s), v64);
      next_claim();
      publish_empty();
    }
  }
  __hip_atomic_fetch_add(ctrl + RCTL_ADONE, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // all lanes: + 64
  RL_FLUSH(0);
}

}  // namespace dfx
