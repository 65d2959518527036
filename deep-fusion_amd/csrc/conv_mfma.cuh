// conv_mfma.cuh -- fused u8 x s8 conv3x3 (stride 1) + ReLU + requant + conv1x1
// (+ReLU) + requant as two chained int8-MFMA implicit GEMMs (gfx950 / CDNA4).
//
// Replaces the reference's JIT micro-kernel and its driver loops:
//   compute_loop / store_output        /root/reference/src/jit_conv_kernel.cc:317-393, :218-305
//   compute1x1_loop / store_1x1output  src/jit_conv_kernel.cc:143-191, :50-141
//   infer_conv0conv1                   src/op_conv.cc:140-260
//
// Structure
//  * ONE PERSISTENT WORKGROUP PER CU: 16 waves (4 per SIMD, <= 128 VGPRs: a single
//    wave issues at most one VALU instruction every ~4-5 cycles, so the requant
//    epilogue needs the occupancy).  The 3x3 and 1x1 weights, packed in MFMA
//    fragment order by the host, are copied to LDS ONCE per CU.
//  * The 16 waves form two independent TEAMS of 7 compute waves + 1 loader wave.
//    A team works on "units" (TH output rows x TW output columns of one image)
//    drawn from a device-side queue (one atomicAdd per unit: the balance211 of
//    op_conv.cc:155-156 made dynamic).  Each team owns TWO input-tile buffers in
//    LDS.  The loader holds the whole halo tile of the next unit in its registers
//    (global loads issued early), writes it (xor 0x80, swizzled) into the free
//    buffer and publishes it with an LDS flag; compute waves consume a buffer and
//    count themselves off on an LDS counter.  There is no workgroup barrier after
//    start-up: compute waves never wait for a global load, for the loader's LDS
//    write, or for each other, so the MFMA, VALU and store phases of different
//    waves interleave instead of running in lock step.
//  * conv0 is D0[oc][px] = sum_k W0[oc][k] * X[k][px] with
//    v_mfma_i32_32x32x32_i8: packed s8 weights are the A operand (rows = oc),
//    input pixels the B operand (columns = px).  One MFMA eats 32 input channels
//    of one (kh,kw) tap.  Both operands come from LDS with ds_read_b128: the
//    weight image is lane-linear, the input halo tile is [row][col][ic] with a
//    16-byte-chunk XOR swizzle that makes the 64 B/pixel stride conflict free.
//  * MFMA i8 is signed x signed.  Activations sit in LDS as (u8 xor 0x80) =
//    u8 - 128, the MFMA chain starts from the inline constant 0, and the
//    compensation comp0[oc] = 128 * sum_k W0[oc][k] is added as an f32 right after
//    the int->f32 conversion: both |raw acc| and |comp| are < 2^24 here (K <= 576),
//    hence exactly representable, and the single f32 add of two exact values is
//    the correctly rounded sum = exactly what vcvtdq2ps gives on the true s32
//    accumulator.  Zero padding is the byte 0x80 (= real 0), which keeps one
//    compensation constant valid at the borders.
//  * After conv0 a lane holds, for its pixel, 16 accumulators per 32-oc block at
//    oc = 32r + 8q + 4h + i (h = lane>>5).  They are requantised in registers
//    (ReLU, scale, round, saturate to u8), packed 4 per dword, and those 16 bytes
//    per block ARE the A fragment of the 1x1 MFMA because the host packed the
//    1x1 weights in exactly this k order: the intermediate activation never
//    leaves the register file (the reference keeps it in xmm registers,
//    jit_conv_kernel.cc:275-277).
//  * conv1 is D1[px][oc1] = sum_oc mid[px][oc] * W1[oc][oc1]: lane = output
//    channel, registers = pixels, so bias/scale are per-lane constants.  The host
//    also permutes which channel each MFMA column computes: within a group of G
//    column blocks lane L owns channels 32G*cg + G*L + {0..G-1}, so every pixel
//    is written with one G*4-byte (s32/f32) or G-byte (s8/u8) store per lane
//    and a half-wave writes 128*G (or 32*G) contiguous bytes: whole HBM lines.
//  * Requantisation has two code paths selected by a wave-uniform flag the host
//    sets: "fast" (both round modes nearest-even, and the host proved from the
//    weights that no value can reach +-2^31 or be NaN, so the x86 overflow/NaN
//    selects are dead) and "exact" (everything else).  Both are bit-identical
//    to the reference arithmetic on the inputs they accept.
//
// Supported here: kh = kw = 3, stride 1, pad in {0,1}, ic/oc in {32,64}, oc1x1 a
// multiple of 32.  Everything else goes to conv_generic.hip.
#pragma once

#include <type_traits>

#include "dfx_device.cuh"

namespace dfx {

constexpr int MFMA_THREADS = 1024;  // 16 waves = 2 teams x (7 compute + 1 loader)
constexpr int MFMA_TEAMS = 2;
constexpr int MFMA_CW = 7;          // compute waves per team
constexpr int MFMA_CTRL_BYTES = 64; // LDS control block: per team full[2], done[2], unit[2]
constexpr int MFMA_LC = 22;        // 16-byte chunks the loader wave holds per lane (88 VGPRs)

__device__ __forceinline__ v16i mfma_i8(v4i a, v4i b, v16i c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

template <int CP>  // 16-byte chunks per pixel; chunk j of LDS pixel P sits at j ^ swz(P)
__device__ __forceinline__ int chunk_swizzle(int P) {
  if (CP == 2) return (P >> 3) & 1;
  if (CP == 4) return (P >> 2) & 3;
  return (P >> 1) & 7;  // CP == 8
}

struct MfmaGeom {  // unit decomposition chosen by the host (dfx_api.hip)
  int th, tw;      // unit size in output rows / columns
  int uy, ux;      // units per image along y / x
  int linear;      // 1: tw == ow, pixels of a unit are numbered linearly across rows
                   // 0: tw % 32 == 0, every 32-pixel tile lies inside one row
  int total_units;
  int row_chunks;   // (tw + 2) * (ic / 16)
  int tile_chunks;  // (th + 2) * row_chunks
  unsigned row_magic;  // ceil(2^32 / row_chunks): q / row_chunks == umulhi(q, row_magic)
  int fast;            // 1: requant fast path is valid (see header comment)
  int tile_stride;     // bytes between the 2 x MFMA_TEAMS input-tile buffers in LDS
  int static_rounds;   // units a team owns statically before it turns to the queue
  int *queue;          // [0] next unit, [1] finished loaders; both 0 between launches
#ifdef DFX_STAMPS
  unsigned long long *prof;  // diagnostic build only: [workgroup][wave][16] cycle sums
#endif
};

#ifdef DFX_STAMPS
// In-kernel stamps (diagnostic build only; never quote this build's run time).
__device__ __forceinline__ unsigned long long dfx_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define DFX_STAMP(var) const unsigned long long var = dfx_stamp()
#define DFX_ACC(slot, expr) prof_acc[slot] += (expr)
#else
#define DFX_STAMP(var)
#define DFX_ACC(slot, expr)
#endif

// f32 value of an accumulator before scaling: vcvtdq2ps(acc) + bias, with the
// u8->s8 compensation folded in as an exact f32 add (see header comment)
__device__ __forceinline__ float acc_to_f32(int raw, float comp, float bias) {
  return __fadd_rn(__fadd_rn(__int2float_rn(raw), comp), bias);
}

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef DFX_RING
#define DFX_RING 4  // conv0 fragment ring depth (see the conv0 loop)
#endif

// The output is written once and never re-read by this kernel: non-temporal stores
// keep it from displacing the input rows / weights in L2 and leave fewer dirty lines
// to write back at the end of the kernel.
#ifndef DFX_TEMPORAL_STORES
#define DFX_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define DFX_STORE(ptr, val) (*(ptr) = (val))
#endif

// ---- one pixel's G consecutive channels -> one store.
// FAST (host-proven preconditions, dfx_api.hip): both stages round to nearest-even;
// every value is finite and |f| < 2^31, so the x86 overflow/NaN selects are dead;
// comp + bias is an exact integer-valued f32 and |acc + bias| < 2^24, so the single
// add of cb = comp + bias equals the reference's float(acc) + float(bias) bit for
// bit.  u8 output then uses v_cvt_pk_u8_f32 (RNE + [0,255] saturation, probed on
// gfx950: tools/probe/probe_valu.hip), which also subsumes the ReLU.
// EXACT: everything else (any round mode, x86 overflow/NaN semantics, two adds). ----
template <int DST, int G, bool FAST>
__device__ __forceinline__ void store_group(unsigned char *p, const int (&acc)[G],
                                            const float (&cp)[G], const float (&bs)[G],
                                            const float (&sc)[G], bool relu, int rm) {
  float f[G];
  if (FAST) {  // bs holds comp + bias; packed f32 math where G allows (v_pk_add/mul_f32)
    if (G >= 2) {
#pragma unroll
      for (int c = 0; c < G; c += 2) {
        v2f x = {__int2float_rn(acc[c]), __int2float_rn(acc[c + 1])};
        x = (x + v2f{bs[c], bs[c + 1]}) * v2f{sc[c], sc[c + 1]};
        f[c] = x[0];
        f[c + 1] = x[1];
      }
    } else {
      f[0] = __fmul_rn(__fadd_rn(__int2float_rn(acc[0]), bs[0]), sc[0]);
    }
  } else {
#pragma unroll
    for (int c = 0; c < G; ++c) f[c] = __fmul_rn(acc_to_f32(acc[c], cp[c], bs[c]), sc[c]);
  }
  if (DST == DFX_F32) {
#pragma unroll
    for (int c = 0; c < G; ++c) f[c] = relu ? relu_x86(f[c]) : f[c];
    if (G == 4) DFX_STORE(reinterpret_cast<v4f *>(p), (v4f{f[0], f[1], f[2], f[3]}));
    else if (G == 2) *reinterpret_cast<float2 *>(p) = float2{f[0], f[1]};
    else *reinterpret_cast<float *>(p) = f[0];
  } else if (DST == DFX_S32) {
    int v[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      if (FAST) v[c] = (int)__builtin_rintf(relu ? __builtin_fmaxf(f[c], 0.0f) : f[c]);
      else v[c] = cvt_x86_rt(relu ? relu_x86(f[c]) : f[c], rm);
    }
    if (G == 4) DFX_STORE(reinterpret_cast<v4i *>(p), (v4i{v[0], v[1], v[2], v[3]}));
    else if (G == 2) *reinterpret_cast<int2 *>(p) = int2{v[0], v[1]};
    else *reinterpret_cast<int *>(p) = v[0];
  } else {
    unsigned pk = 0;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      if (FAST && DST == DFX_U8) {
        pk = __builtin_amdgcn_cvt_pk_u8_f32(f[c], c, pk);
      } else {
        const float fr = relu ? (FAST ? __builtin_fmaxf(f[c], 0.0f) : relu_x86(f[c])) : f[c];
        const int v = FAST ? (int)__builtin_rintf(fr) : cvt_x86_rt(fr, rm);
        const unsigned b = (DST == DFX_U8) ? sat_u8_bits(v) : ((unsigned)sat_s8(v) & 0xffu);
        pk |= b << (8 * c);
      }
    }
    if (G == 4) DFX_STORE(reinterpret_cast<unsigned *>(p), pk);
    else if (G == 2) *reinterpret_cast<unsigned short *>(p) = (unsigned short)pk;
    else *p = (uint8_t)pk;
  }
}

// 1-byte outputs: the same arithmetic as store_group, returning the G packed bytes instead
// of storing them (conv_stream.cuh stages them in LDS to write 16 bytes per lane).
template <int DST, int G, bool FAST>
__device__ __forceinline__ unsigned pack_group(const int (&acc)[G], const float (&cp)[G], const float (&bs)[G],
                                               const float (&sc)[G], bool relu, int rm) {
  static_assert(DST == DFX_U8 || DST == DFX_S8, "1-byte outputs only");
  unsigned pk = 0;
#pragma unroll
  for (int c = 0; c < G; ++c) {
    float f;
    if (FAST) f = __fmul_rn(__fadd_rn(__int2float_rn(acc[c]), bs[c]), sc[c]);
    else f = __fmul_rn(acc_to_f32(acc[c], cp[c], bs[c]), sc[c]);
    if (FAST && DST == DFX_U8) {
      pk = __builtin_amdgcn_cvt_pk_u8_f32(f, c, pk);
    } else {
      const float fr = relu ? (FAST ? __builtin_fmaxf(f, 0.0f) : relu_x86(f)) : f;
      const int v = FAST ? (int)__builtin_rintf(fr) : cvt_x86_rt(fr, rm);
      const unsigned b = (DST == DFX_U8) ? sat_u8_bits(v) : ((unsigned)sat_s8(v) & 0xffu);
      pk |= b << (8 * c);
    }
  }
  return pk;
}

// FUSED = false is the unfused conv() overload (reference deepfusion.h:121-129): the same
// loader / tile machinery and 3x3 MFMA ring, but the contraction is oriented
// D0[px][oc] (A = input pixels, B = weights packed with the channel permutation, G ==
// OCB) so that lane = output channel and the typed store is coalesced like the 1x1 stage.
template <int ICB, int OCB, int G, int DST, bool FUSED = true>
__global__ __launch_bounds__(MFMA_THREADS, 4) void conv_mfma_fused_kernel(ConvArgs a, MfmaGeom g) {
  constexpr int IC = 32 * ICB, OC = 32 * OCB, CP = IC / 16;
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  DFX_STAMP(t_entry);
#ifdef DFX_STAMPS
  unsigned long long rt_entry;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_entry)::"memory");
#endif
  const int OC1 = FUSED ? a.oc1 : 0, NCB = OC1 >> 5, NCG = FUSED ? NCB / G : 1;
  unsigned char *w0s = smem;                                   // [OCB][9][ICB][64 lanes][16 B]
  unsigned char *w1s = w0s + OCB * 9 * ICB * 1024;             // [NCB][OCB][64 lanes][16 B]
  float *cst = reinterpret_cast<float *>(w1s + NCB * OCB * 1024);
  const int cst_bytes = (3 * (OC + OC1) * 4 + 15) & ~15;
  int *ctrl = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(cst) + cst_bytes);
  unsigned char *tiles = reinterpret_cast<unsigned char *>(ctrl) + MFMA_CTRL_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int team = wave >> 3, cw = wave & 7;                  // cw == MFMA_CW: the team's loader
  const int LW = g.tw + 2;
  const int upi = g.uy * g.ux;
  // per-team control words (LDS): full[b] = number of tiles published into buffer b,
  // done[b] = number of compute-wave completions on buffer b, unit[b] = unit id or -1
  int *full = ctrl + team * 6, *done = full + 2, *unit_of = full + 4;
  unsigned char *team_tiles = tiles + (size_t)team * 2 * g.tile_stride;

  // ---- halo-tile chunk helpers (loader waves; compute waves for the very first tile) ----
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
  // chunk q of a unit's halo tile: LDS row lr = q / row_chunks, chunk c within the row
  auto load_chunk = [&](const uint8_t *src_n, int y0, int x0, int q) {
    const int lr = (int)__umulhi((unsigned)q, g.row_magic);
    const int c = q - lr * g.row_chunks;
    const int iy = y0 + lr, ix = x0 + c / CP;
    const bool ok = q < g.tile_chunks && iy >= 0 && iy < a.ih && ix >= 0 && ix < a.iw;
    // branch-free: out-of-image chunks read offset 0 of the image and are zeroed
    // (real 0).  32-bit offset off a uniform base.
    const unsigned off = ok ? (unsigned)((iy * a.iw + ix) * IC + 16 * (c % CP)) : 0u;
    const v4i v = *reinterpret_cast<const v4i *>(src_n + off);
    return (ok ? v : v4i{0, 0, 0, 0}) ^ x80;  // stored form: u8 - 128; padding = 0x80
  };
  // LDS byte offset of chunk q (unit independent); chunks beyond the tile go to a
  // 16-byte dump slot right behind it so the write loop needs no predicate
  auto chunk_lds_off = [&](int q) {
    const int lr = (int)__umulhi((unsigned)q, g.row_magic);
    const int c = q - lr * g.row_chunks;
    const int P = lr * LW + c / CP;
    return q < g.tile_chunks ? P * IC + 16 * ((c % CP) ^ chunk_swizzle<CP>(P)) : g.tile_chunks * 16;
  };
  auto unit_origin = [&](int unit, const uint8_t *&src_n, int &y0, int &x0) {
    const int n = unit / upi, u = unit - n * upi;
    const int uyi = u / g.ux, uxi = u - uyi * g.ux;
    y0 = uyi * g.th - a.pt;
    x0 = uxi * g.tw - a.pl;
    src_n = a.src + (size_t)n * a.ih * a.iw * IC;
  };

  // The FIRST tile of each team is staged cooperatively by the team's 7 compute waves right
  // after the weights (2-3 chunks per thread, one memory round trip), and published before
  // the barrier: the loader's own start-up (per-lane staging table, 22-chunk prefetch address
  // math: ~11 k cycles, measured) used to sit between the barrier and the first MFMA.  The
  // loader then starts with the team's second unit.  Needs a statically known first unit.
  const bool coop0 = g.static_rounds >= 1;

  // ---- weights + constants: the host keeps them in ONE device buffer laid out
  //      exactly like the LDS image [W0 fragments | W1 fragments | constants], so a
  //      single linear copy stages them.  All global loads of a pass are issued
  //      before the first LDS write: one memory round trip per 128 KB. ----
  auto stage_weights = [&]() {  // called by the 14 compute waves (the loaders hold tile data)
    constexpr int NT = MFMA_TEAMS * MFMA_CW * 64;
    constexpr int TT = MFMA_CW * 64;  // threads of one team's compute waves
    const int ctid = (team * MFMA_CW + cw) * 64 + lane, tctid = cw * 64 + lane;
    const v4i *s = reinterpret_cast<const v4i *>(a.wei);
    v4i *d = reinterpret_cast<v4i *>(smem);
    const int total = OCB * 9 * ICB * 64 + NCB * OCB * 64 + (3 * (OC + OC1) * 4 + 15) / 16;
    // the team's first tile (coop0): buffer 0, unit = team id; its first 4 chunks per thread
    // travel together with the weights (one memory round trip for both)
    const int unit0 = (int)blockIdx.x * MFMA_TEAMS + team;
    const bool tile0 = coop0 && unit0 < g.total_units;
    const uint8_t *src_n = a.src;
    int y0 = 0, x0 = 0;
    if (tile0) unit_origin(unit0, src_n, y0, x0);
    for (int base = 0; base < total; base += 4 * NT) {
      const int q0 = base + ctid, last = total - 1;
      const v4i t0 = s[min(q0 + 0 * NT, last)];
      const v4i t1 = s[min(q0 + 1 * NT, last)];
      const v4i t2 = s[min(q0 + 2 * NT, last)];
      const v4i t3 = s[min(q0 + 3 * NT, last)];
      v4i u0 = x80, u1 = x80, u2 = x80, u3 = x80;
      if (tile0 && base == 0) {
        u0 = load_chunk(src_n, y0, x0, tctid + 0 * TT);
        u1 = load_chunk(src_n, y0, x0, tctid + 1 * TT);
        u2 = load_chunk(src_n, y0, x0, tctid + 2 * TT);
        u3 = load_chunk(src_n, y0, x0, tctid + 3 * TT);
      }
      d[min(q0 + 0 * NT, last)] = t0;
      d[min(q0 + 1 * NT, last)] = t1;
      d[min(q0 + 2 * NT, last)] = t2;
      d[min(q0 + 3 * NT, last)] = t3;
      if (tile0 && base == 0) {
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(tctid + 0 * TT)) = u0;
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(tctid + 1 * TT)) = u1;
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(tctid + 2 * TT)) = u2;
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(tctid + 3 * TT)) = u3;
      }
    }
    if (tile0) {  // a tile of more than 4 chunks per thread: the rest
      for (int base = 4 * TT; base < g.tile_chunks; base += 4 * TT) {
        const int q0 = base + tctid;
        const v4i t0 = load_chunk(src_n, y0, x0, q0 + 0 * TT);
        const v4i t1 = load_chunk(src_n, y0, x0, q0 + 1 * TT);
        const v4i t2 = load_chunk(src_n, y0, x0, q0 + 2 * TT);
        const v4i t3 = load_chunk(src_n, y0, x0, q0 + 3 * TT);
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(q0 + 0 * TT)) = t0;
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(q0 + 1 * TT)) = t1;
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(q0 + 2 * TT)) = t2;
        *reinterpret_cast<v4i *>(team_tiles + chunk_lds_off(q0 + 3 * TT)) = t3;
      }
    }
    if (ctid < MFMA_CTRL_BYTES / 4) {
      // control block: zero, except that with a cooperatively staged first tile (coop0 below)
      // buffer 0 of each team starts out published: full[0] = 1, unit[0] = the team's unit 0
      int v = 0;
      const int ti = ctid / 6, f = ctid - 6 * ti;
      if (g.static_rounds >= 1 && ti < MFMA_TEAMS) {
        const int u0 = (int)blockIdx.x * MFMA_TEAMS + ti;
        if (f == 0) v = 1;
        if (f == 4) v = u0 < g.total_units ? u0 : -1;
      }
      ctrl[ctid] = v;
    }
  };

  if (cw == MFMA_CW) {
    // =========================== loader wave ===========================
    v4i pf[MFMA_LC];
    // issue-early half: first 64*MFMA_LC chunks of a unit -> registers
#define DFX_PREFETCH(UNIT)                                                              \
  do {                                                                                  \
    const uint8_t *src_n_;                                                              \
    int y0_, x0_;                                                                       \
    unit_origin((UNIT), src_n_, y0_, x0_);                                              \
    int lq_ = lane; /* opaque: keep the per-chunk index math out of the LICM set */     \
    asm volatile("" : "+v"(lq_));                                                       \
    _Pragma("unroll") for (int i = 0; i < MFMA_LC; ++i) pf[i] =                         \
        load_chunk(src_n_, y0_, x0_, lq_ + 64 * i);                                     \
  } while (0)
    // Unit sequence of this team: the first `static_rounds` units are owned
    // statically (round j -> unit j*T + team id; no atomic: 2 x gridDim loaders
    // hammering one queue word at kernel start cost ~12 us), the rest come from the
    // device-side queue.  Draws are device-scope atomics that take microseconds to
    // return: one is kept in flight and only broadcast (readfirstlane = wait) when needed.
    const int T = (int)gridDim.x * MFMA_TEAMS, tg = (int)blockIdx.x * MFMA_TEAMS + team;
    auto unit_at = [&](int j) {
      int v = j * T + tg;
      if (j >= g.static_rounds) {
        v = 0x7fffffff;
        if (lane == 0) v = g.static_rounds * T + atomicAdd(g.queue, 1);
      }
      return v;
    };

    // coop0: the compute waves stage the team's first tile themselves; the loader passes
    // the barrier at once (nobody waits for its start-up) and begins with the second unit
    DFX_STAMP(l_pre);
    if (coop0) __syncthreads();
    const int j0 = coop0 ? 1 : 0;
    int cur = __builtin_amdgcn_readfirstlane(unit_at(j0));
    int nxt_v = unit_at(j0 + 1);
    int jn = j0 + 2;
    if (cur < g.total_units) DFX_PREFETCH(cur);  // (!coop0: the first tile's loads fly during the weight copy)
    int wr_off[MFMA_LC];  // (computed while those loads are in flight)
#pragma unroll
    for (int i = 0; i < MFMA_LC; ++i) wr_off[i] = chunk_lds_off(lane + 64 * i);
    if (!coop0) __syncthreads();  // the only workgroup barrier: weights + control block are in LDS
    DFX_STAMP(l_post);
#ifdef DFX_STAMPS
    if (lane == 0) {  // loader: cycles from entry to its first loop iteration
      unsigned long long *o = g.prof + ((size_t)blockIdx.x * 16 + wave) * 16;
      o[13] = l_pre - t_entry;
      o[14] = l_post - l_pre;
    }
#endif

    for (int k = j0;; ++k) {
      const int b = k & 1;
      unsigned char *ins = team_tiles + (size_t)b * g.tile_stride;
      // buffer b is free once the 7 compute waves have finished its previous tile
      while (__hip_atomic_load(done + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < MFMA_CW * (k >> 1))
        __builtin_amdgcn_s_sleep(2);
      const bool valid = cur < g.total_units;
      if (valid) {
#pragma unroll
        for (int i = 0; i < MFMA_LC; ++i)   // write-late half of the staging
          *reinterpret_cast<v4i *>(ins + wr_off[i]) = pf[i];
        if (g.tile_chunks > 64 * MFMA_LC) {  // oversized tile: the rest is staged synchronously
          const uint8_t *src_n; int y0, x0;
          unit_origin(cur, src_n, y0, x0);
          for (int q = 64 * MFMA_LC + lane; q < g.tile_chunks; q += 64)
            *reinterpret_cast<v4i *>(ins + chunk_lds_off(q)) = load_chunk(src_n, y0, x0, q);
        }
      }
      if (lane == 0) unit_of[b] = valid ? cur : -1;
      // publish: LDS executes a wave's DS instructions in order; the release makes the
      // compiler keep them ahead of the flag store and waits for them to complete
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // all lanes' tile writes first
      if (lane == 0) __hip_atomic_store(full + b, (k >> 1) + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!valid) break;
      cur = __builtin_amdgcn_readfirstlane(nxt_v);
      nxt_v = unit_at(jn++);
      if (cur < g.total_units) DFX_PREFETCH(cur);
    }
#undef DFX_PREFETCH
    // last loader out re-arms the queue for the next launch.  The draw still in
    // flight must have been performed before this loader counts itself out, or it
    // could land after the reset.
    const int pending = __builtin_amdgcn_readfirstlane(nxt_v);
    if (lane == 0 && pending >= 0) {
      const int fin = atomicAdd(g.queue + 1, 1);
      if (fin == (int)gridDim.x * MFMA_TEAMS - 1) {
        atomicExch(g.queue, 0);
        atomicExch(g.queue + 1, 0);
      }
    }
    return;
  }

  // =========================== compute waves ===========================
  stage_weights();
  DFX_STAMP(t_staged);  // (diagnostic builds: the stamp waits for this wave's LDS writes)
  __syncthreads();
  const int l31 = lane & 31, h = lane >> 5;
  const float *comp0 = cst, *bias0 = cst + OC, *scale0 = cst + 2 * OC;
  const float *comp1 = cst + 3 * OC, *bias1 = cst + 3 * OC + OC1, *scale1 = cst + 3 * OC + 2 * OC1;
  const bool relu1 = (FUSED ? a.relu1 : a.relu0) || DST == DFX_U8;  // ReLU of the stage that stores
  const bool fast = g.fast != 0;
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned row_bytes = (unsigned)(FUSED ? OC1 : OC) * ESZ;  // dst bytes per pixel

#ifdef DFX_STAMPS
  unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DFX_STAMP(t_loop);
  DFX_ACC(4, t_loop - t_entry);  // start-up: weights staging + barrier
#ifdef DFX_STAMPS
  const unsigned long long startup_stage = t_staged - t_entry;
#endif
  int rot = 0;  // tile i of a unit goes to compute wave (rot + i) % MFMA_CW; rot advances by the
               // unit's tile count, so units with fewer than MFMA_CW tiles keep every wave busy
  for (int k = 0;; ++k) {
    const int b = k & 1;
    const unsigned char *ins = team_tiles + (size_t)b * g.tile_stride;
    DFX_STAMP(c0);
    while (__hip_atomic_load(full + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (k >> 1) + 1)
      __builtin_amdgcn_s_sleep(1);
    DFX_STAMP(c1);
    DFX_ACC(0, c1 - c0);  // wait for the tile
    const int unit = __builtin_amdgcn_readfirstlane(unit_of[b]);
    if (unit < 0) break;

    const int n = unit / upi, u = unit - n * upi;
    const int uyi = u / g.ux, uxi = u - uyi * g.ux;
    const int y0 = uyi * g.th, x0 = uxi * g.tw;
    const int th = min(g.th, a.oh - y0), tw = min(g.tw, a.ow - x0);
    const int npx = th * tw;
    const int tiles_per_row = (tw + 31) >> 5;
    const int ntiles = g.linear ? (npx + 31) >> 5 : th * tiles_per_row;

    int t0 = cw - rot;
    if (t0 < 0) t0 += MFMA_CW;
    rot = (rot + ntiles) % MFMA_CW;
    for (int t = t0; t < ntiles; t += MFMA_CW) {
      // pixel of this lane (conv0 column) and the tile's output base (wave-uniform)
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

      DFX_STAMP(c2);
      // ---- conv0: 9 taps x ICB k-steps x OCB row blocks ----
      // Explicit RD-deep fragment ring: the fragments of k-step s+RD-1 are fetched right
      // after the MFMAs of step s were issued, into the registers last read by the
      // MFMAs of step s-1.  An LDS load must never target a register that a just-issued
      // MFMA still has to read as A/B operand (see the note at the 1x1 stage); here at
      // least OCB MFMAs separate the two.  sched_barrier keeps hipcc from re-mixing.
      v16i acc0[OCB];
#pragma unroll
      for (int r = 0; r < OCB; ++r) acc0[r] = zero16;
      {
        constexpr int NS = 9 * ICB;  // k-steps
        constexpr int RD = DFX_RING;  // ring slots; fragments are fetched RD-1 steps ahead
        v4i fb[RD], fw[RD][OCB];
        auto fetch = [&](int st, int slot) {  // st, slot are compile-time after unrolling
          const int tap = st / ICB, c = st % ICB;
          const int P = Pb + (tap / 3) * LW + (tap % 3);
          fb[slot] = *reinterpret_cast<const v4i *>(ins + P * IC + 16 * ((2 * c + h) ^ chunk_swizzle<CP>(P)));
#pragma unroll
          for (int r = 0; r < OCB; ++r)
            fw[slot][r] = *reinterpret_cast<const v4i *>(w0s + ((r * 9 + tap) * ICB + c) * 1024 + lane16);
        };
#pragma unroll
        for (int st = 0; st < RD - 1; ++st) fetch(st, st);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < OCB; ++r)
            acc0[r] = FUSED ? mfma_i8(fw[st % RD][r], fb[st % RD], acc0[r])   // D0[oc][px]
                            : mfma_i8(fb[st % RD], fw[st % RD][r], acc0[r]);  // D0[px][oc]
          __builtin_amdgcn_sched_barrier(0);
          if (st + RD - 1 < NS) fetch(st + RD - 1, (st + RD - 1) % RD);
        }
      }

      unsigned char *tile_dst = reinterpret_cast<unsigned char *>(a.dst) + obase * row_bytes;
      using T = std::true_type;
      using F = std::false_type;
      if constexpr (!FUSED) {
        // ---- unfused: requant 0 + typed store straight from the 3x3 accumulators ----
        const int chb = lch;  // G == OCB: this lane owns channels G*l31 .. G*l31 + G-1
        float cp[G], bs[G], sc[G];
#pragma unroll
        for (int cc = 0; cc < G; ++cc) {
          cp[cc] = comp0[chb + cc];
          bs[cc] = bias0[chb + cc];
          sc[cc] = scale0[chb + cc];
        }
        const unsigned lane_off = (unsigned)h4 * row_bytes + (unsigned)chb * ESZ;
        auto emit0 = [&](auto fast_tag, auto check_tag) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int pl = 8 * (e >> 2) + (e & 3);  // + 4h, folded into lane_off
            if (!decltype(check_tag)::value || pl + 4 * h < nvalid) {
              int v[G];
#pragma unroll
              for (int cc = 0; cc < G; ++cc) v[cc] = acc0[cc][e];
              store_group<DST, G, decltype(fast_tag)::value>(
                  (tile_dst + (size_t)((unsigned)pl * row_bytes)) + lane_off, v, cp, bs, sc, relu1, a.rm0);
            }
          }
        };
        if (fast) { if (nvalid == 32) emit0(T{}, F{}); else emit0(T{}, T{}); }
        else      { if (nvalid == 32) emit0(F{}, F{}); else emit0(F{}, T{}); }
      } else {
      DFX_STAMP(c3);
      DFX_ACC(1, c3 - c2);  // conv0 MFMA issue
      // ---- requant 0 in registers -> A fragments of the 1x1 ----
      v4i mid[OCB];
      if (fast) {  // cb0 = comp0 + bias0 (exact, host-proven); ReLU + RNE + sat + pack in one op
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ch = 32 * r + 8 * q + h4;
            const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
            const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              v2f x = {__int2float_rn(acc0[r][4 * q + i]), __int2float_rn(acc0[r][4 * q + i + 1])};
              x = (x + v2f{bs[i], bs[i + 1]}) * v2f{sc[i], sc[i + 1]};
              pk = __builtin_amdgcn_cvt_pk_u8_f32(x[0], i, pk);
              pk = __builtin_amdgcn_cvt_pk_u8_f32(x[1], i + 1, pk);
            }
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      } else {
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ch = 32 * r + 8 * q + h4;
            const v4f cp = *reinterpret_cast<const v4f *>(comp0 + ch);
            const v4f bs = *reinterpret_cast<const v4f *>(bias0 + ch);
            const v4f sc = *reinterpret_cast<const v4f *>(scale0 + ch);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float f = __fmul_rn(acc_to_f32(acc0[r][4 * q + i], cp[i], bs[i]), sc[i]);
              pk |= sat_u8_bits(cvt_x86_rt(relu_x86(f), a.rm0)) << (8 * i);
            }
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      }
      DFX_STAMP(c4);
      DFX_ACC(2, c4 - c3);  // requant 0
      // ---- conv1 + requant 1 + store, G column blocks at a time ----
      for (int cg = 0; cg < NCG; ++cg) {
        const int chb = 32 * G * cg + lch;  // this lane's first channel in the group
        v16i acc1[G];
        float cp[G], bs[G], sc[G];
#pragma unroll
        for (int cc = 0; cc < G; ++cc) {
          acc1[cc] = zero16;
          cp[cc] = comp1[chb + cc];
          bs[cc] = bias1[chb + cc];
          sc[cc] = scale1[chb + cc];
        }
        // All W1 fragments of the group are fetched into DISTINCT registers before the
        // MFMA chain starts, and the chain is fenced off from the loads: no register
        // that an in-flight MFMA reads as A/B is the destination of a later LDS load.
        // (With the fragments rotating through two register quads, back-to-back MFMAs
        // followed by a reload of the second one's B operand produced rare wrong
        // accumulators on the first launches of a process on gfx950.)
        v4i wf[OCB][G];
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int cc = 0; cc < G; ++cc)
            wf[r][cc] = *reinterpret_cast<const v4i *>(
                w1s + ((cg * G + cc) * OCB + r) * 1024 + lane16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int cc = 0; cc < G; ++cc) acc1[cc] = mfma_i8(mid[r], wf[r][cc], acc1[cc]);
        __builtin_amdgcn_sched_barrier(0);
        // register e of the accumulator = pixel 8*(e>>2) + 4h + (e&3) of the tile
        const unsigned lane_off = (unsigned)h4 * row_bytes + (unsigned)chb * ESZ;
        auto emit = [&](auto fast_tag, auto check_tag) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int pl = 8 * (e >> 2) + (e & 3);  // + 4h, folded into lane_off
            if (!decltype(check_tag)::value || pl + 4 * h < nvalid) {
              int v[G];
#pragma unroll
              for (int cc = 0; cc < G; ++cc) v[cc] = acc1[cc][e];
              store_group<DST, G, decltype(fast_tag)::value>(
                  (tile_dst + (size_t)((unsigned)pl * row_bytes)) + lane_off, v, cp, bs, sc, relu1, a.rm1);
            }
          }
        };
        if (fast) { if (nvalid == 32) emit(T{}, F{}); else emit(T{}, T{}); }
        else      { if (nvalid == 32) emit(F{}, F{}); else emit(F{}, T{}); }
      }
      DFX_STAMP(c5);
      DFX_ACC(3, c5 - c4);  // conv1 + requant 1 + stores
      }  // FUSED
      DFX_ACC(6, 1);
    }
    DFX_STAMP(c6);
    // this wave is done with buffer b (its LDS reads have been consumed by the MFMAs)
    // (one add per WAVE: lane 0 only)
    if (lane == 0) __hip_atomic_fetch_add(done + b, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    DFX_ACC(5, c6 - c0);  // whole unit
    DFX_ACC(7, 1);
  }
#ifdef DFX_STAMPS
  {
    DFX_STAMP(t_end);
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 0) {
      unsigned long long *o = g.prof + ((size_t)blockIdx.x * 16 + wave) * 16;
      for (int k = 0; k < 8; ++k) o[k] = prof_acc[k];
      o[8] = t_entry; o[9] = t_end; o[10] = rt; o[11] = rt_entry; o[12] = startup_stage;
    }
  }
#endif
}

}  // namespace dfx
