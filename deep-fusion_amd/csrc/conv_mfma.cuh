// conv_mfma.cuh -- fused u8 x s8 conv3x3 (stride 1) + ReLU + requant + conv1x1
// (+ReLU) + requant as two chained int8-MFMA implicit GEMMs (gfx950 / CDNA4).
//
// Replaces the reference's JIT micro-kernel and its driver loops:
//   compute_loop / store_output        /root/reference/src/jit_conv_kernel.cc:317-393, :218-305
//   compute1x1_loop / store_1x1output  src/jit_conv_kernel.cc:143-191, :50-141
//   infer_conv0conv1                   src/op_conv.cc:140-260
//
// Structure
//  * ONE PERSISTENT WORKGROUP PER CU: 16 waves (4 per SIMD, <= 128 VGPRs) = 14 compute
//    waves + 2 loader waves.  One wave issues at most one vector instruction every
//    ~4.5 cycles (v_cvt_f32_i32: 7.5; tools/probe/probe_valu3.hip), a SIMD retires one
//    per ~1.2 cycles: the kernel is bound by per-wave instruction issue, so the design
//    minimises instructions per MFMA and keeps four waves per SIMD.
//    The 3x3 and 1x1 weights, packed in MFMA fragment order by the host, are copied to LDS
//    ONCE per CU.
//  * Work = "units" (TH output rows x TW output columns of one image).  A CU walks a
//    sequence of units k = 0, 1, ...: loader wave L owns the units with k % 2 == L (the
//    first `static_rounds` of them statically, the rest from a device-side queue: the
//    balance211 of op_conv.cc:155-156 made dynamic), stages the unit's input halo tile into
//    ring slot k % 4 of LDS (global loads issued one unit ahead, held in registers,
//    written as u8 xor 0x80 with a column swizzle) and publishes it with an LDS flag.
//  * The 32-pixel tiles of the published units are CLAIMED by the compute waves from one
//    LDS counter (one returning ds_add per tile): any wave takes the next tile of the CU,
//    so the two loaders' streams, partial units and waves on differently loaded SIMDs
//    balance themselves; a slot is free again when all of its tile claims were counted
//    off.  There is no workgroup barrier after start-up.
//  * conv0 is D0[oc][px] = sum_k W0[oc][k] * X[k][px] with v_mfma_i32_32x32x32_i8: packed
//    s8 weights are the A operand (rows = oc), input pixels the B operand (columns = px).
//    One MFMA eats 32 input channels of one (kh,kw) tap.  Both operands come from LDS
//    with ds_read_b128.  The halo tile is [row][col][ic] with the 16-byte chunks of a
//    pixel XOR-swizzled by a function of its COLUMN only, so a lane needs one base
//    address per (tap column, ic half) and every tap row is a wave-uniform offset away:
//    ~30 address instructions per tile instead of ~120.
//  * MFMA i8 is signed x signed.  Activations sit in LDS as (u8 xor 0x80) = u8 - 128 and
//    the exact integer compensation comp[oc] = 128 * sum_k W[oc][k] is folded into the
//    accumulator's initial value (the MFMA's C operand), so the accumulator IS the
//    reference's s32 accumulator.  Zero padding is the byte 0x80 (= real 0).
//  * Requantisation, three host-selected modes per stage (dfx_api.hip proves the
//    preconditions from the actual weights, bias and scales):
//      exact   the x86 instruction chain op for op (vcvtdq2ps, vaddps, vmulps, vmaxps,
//              vcvtps2dq incl. its 0x80000000 overflow/NaN result, vpmovusdb/vpmovsdb)
//      fast    both roundings nearest-even and no value can be NaN or reach +-2^31:
//              v_cvt_f32_i32, one add of (integer-valued) bias, mul, v_cvt_pk_u8_f32
//      magic   additionally |acc + bias| is small enough that the int->f32 conversion
//              can ride on the MFMA for free: the accumulator starts from the BIT PATTERN
//              of a float constant m whose binade has ulp u, so after the integer MACs
//              its bits read as the float m + (acc + bias) * u, exactly.  Subtracting m
//              (one exact add) and multiplying by scale / u (u a power of two, folded into
//              the scale by the host) gives float(acc + bias) * scale with the reference's
//              single rounding, without any v_cvt_f32_i32 -- the most expensive
//              instruction of the chain.  Stage 0 (lane = pixel, register = channel) loads
//              per-channel start values bits(1.5 * 2^23) + comp + bias from LDS; stage 1 and
//              the unfused store stage (lane = channel, registers = pixels) start every
//              accumulator from the inline constant 1/(2*pi) = 0x3E22F983, the one inline
//              float constant whose mantissa is not zero, which costs no register at all.
//    All modes are bit-identical to the oracle on the inputs they accept.
//  * After conv0 a lane holds, for its pixel, 16 accumulators per 32-oc block at
//    oc = 32r + 8q + 4h + i (h = lane>>5).  They are requantised in registers, packed 4 per
//    dword, and those 16 bytes per block ARE the A fragment of the 1x1 MFMA because the
//    host packed the 1x1 weights in exactly this k order: the intermediate activation
//    never leaves the register file (the reference keeps it in xmm registers,
//    jit_conv_kernel.cc:275-277).
//  * conv1 is D1[px][oc1] = sum_oc mid[px][oc] * W1[oc][oc1]: lane = output channel,
//    registers = pixels, so bias/scale are per-lane constants.  The host also permutes
//    which channel each MFMA column computes: within a group of G column blocks lane L
//    owns channels 32G*cg + G*L + {0..G-1}, so every pixel is written with one G*4-byte
//    (s32/f32) or G-byte (s8/u8) store per lane and a half-wave writes 128*G (or 32*G)
//    contiguous bytes: whole HBM lines.
//
//  * What bounds the kernel (round 2, profiles/stamps.py timelines, PMC): issue on the SIMD's vector
//    port, which VALU and MFMA share.  Per 32-pixel tile a wave issues 52 MFMAs (1.7 k matrix-pipe cycles)
//    and ~576 other VALU instructions at 4 cycles each (the u8 epilogue needs two per output value:
//    v_pk_add_f32 + v_pk_mul_f32 on pixel pairs, one v_cvt_pk_u8_f32 per value), and the two pipes
//    mostly alternate (co-execution 17 % of the MFMA-busy cycles).  Hence: no per-pixel address arithmetic in vector
//    registers (scalar pixel bases + one per-lane offset), no int->float conversions (magic start
//    values), lane-derived values recomputed instead of spilled (a scratch reload waits for every store
//    in flight), and the scheduling rules below that keep all four SIMDs fed:
//      - units are split statically and stream-major when a loader gets <= 8 of them, so that every
//        workgroup processes the same number +-1 (the queue's granularity left 6..8 per workgroup);
//      - two of the four ring slots are full when the claim loop starts (the compute waves stage units 0/1
//        with the weights before the only barrier; each loader fetches its second unit right behind it);
//      - a wave draws its next claim after conv0 of the current tile and looks at that tile's unit in
//        the last store group (not a whole tile ahead: parked claims delayed the release of slots);
//      - a wave that lags behind the other 13 raises its issue priority (the arbiter prefers the
//        oldest wave of a SIMD; the youngest took 3x as long per tile and held its unit's slot).
//
// Round-1 note withdrawn: an earlier revision blamed a one-off wrong output on a hardware
// hazard (an LDS load overwriting the A/B registers of an MFMA issued just before it).  The
// isolated probe tools/probe/probe_mfma_war.hip shows no such hazard (0 wrong results in
// 2.6e10 MFMA/load pairs at distance 0..32); see DESIGN.md section 4.1.  The fragment rings
// below are a software pipeline (bounded prefetch depth), nothing more.
//
// Supported here: kh = kw = 3, stride 1, pad in {0,1}, ic/oc in {32,64}, oc1x1 a
// multiple of 32.  Everything else goes to conv_stream.cuh / conv_direct.cuh.
#pragma once

#include <type_traits>

#include "dfx_device.cuh"

namespace dfx {

constexpr int MFMA_THREADS = 1024;  // 16 waves: 14 compute + 2 loaders (waves 7 and 15)
constexpr int MFMA_TEAMS = 2;       // loader waves = unit streams per CU
constexpr int MFMA_CW = 7;          // compute waves per loader
constexpr int MFMA_NB = 4;          // input-tile ring slots in LDS (2 per loader)
#ifndef DFX_STAMPS
constexpr int MFMA_CTRL_BYTES = 128; // LDS control block, see CTL_* below
#else
constexpr int MFMA_CTRL_BYTES = 128 + 16 * 8 * 4; // + the stamps build's per-wave cycle sums ([16 waves][8] ints)
#endif
constexpr int MFMA_LC = 22;         // 16-byte chunks the loader wave holds per lane (88 VGPRs)
constexpr int MFMA_SPIN_LIMIT = 1 << 24;  // bound of every flag wait (~seconds): a protocol error ends the launch
                                          // with wrong output (the parity tests catch it) instead of hanging the GPU

// requant start values (see header): bit patterns the accumulators start from
constexpr int MAGIC0_BITS = 0x4B400000;   // 1.5 * 2^23: ulp 1, room for +-2^22
constexpr float MAGIC0_F = 12582912.0f;
constexpr int MAGIC1_BITS = 0x3E22F983;   // 1/(2*pi), an inline constant: ulp 2^-26, mantissa 0x22F983
constexpr int MAGIC1_LO = -0x22F983;      // most negative / positive integer it can absorb
constexpr int MAGIC1_HI = 0x7FFFFF - 0x22F983;

// Constant area (floats): per conv0 channel A0 | B0 | C0, per 1x1 channel A1 | B1 | C1 (see emit_pair).
// The B and C of the STORING stage (stage 1 of a fused op, stage 0 of an unfused one) are kept as
// PAIRS {k, k}: v_pk_add_f32 / v_pk_mul_f32 then take them as plain 64-bit operands.  Broadcasting
// one 32-bit constant with op_sel instead is not safe on this hardware: the form that takes the HIGH
// half of a source pair for the low lane (op_sel:[0,1]) returned wrong low results in the last 16
// lanes of a wave about once per 1e4 epilogue executions (tools/probe/probe_pk_opsel.hip reproduces it
// in isolation; the form op_sel_hi:[1,0] and scalar arithmetic never did).
__host__ __device__ constexpr int mfma_cst_floats(int oc, int oc1) { return oc1 ? 3 * oc + 5 * oc1 : 5 * oc; }

// control block in LDS (ints)
constexpr int CTL_NEXT = 0;   // 64 x the next tile claim of this CU (every lane of a claiming wave adds 1)
constexpr int CTL_END = 1;    // [2] first k without a unit, per loader (INT_MAX while running)
constexpr int CTL_FULL = 4;   // [4] generations published into slot s
constexpr int CTL_DONE = 8;   // [4] 64 x the tile claims counted off on slot s (cumulative)
constexpr int CTL_INFO = 16;  // [4][4] per slot: dst pixel index of the unit's first pixel, (th << 16) | tw of
                              // the (possibly clipped) unit, ceil(2^32 / tiles per row) or 0, unused --
                              // computed once per unit by the loader so that a tile claim costs no division

__device__ __forceinline__ v16i mfma_i8(v4i a, v4i b, v16i c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

template <int CP>  // 16-byte chunks per pixel; chunk j of the pixel in LDS column X sits at j ^ swz(X)
__device__ __forceinline__ int chunk_swizzle(int X) {
  if (CP == 2) return (X >> 3) & 1;
  if (CP == 4) return (X >> 2) & 3;
  return (X >> 1) & 7;  // CP == 8 (other kernels)
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
  unsigned tw_magic;   // ceil(2^32 / tw)
  unsigned upi_magic, ux_magic;  // ceil(2^32 / (uy * ux)), ceil(2^32 / ux); 0 where the divisor is 1
  int ntu;             // tile claims per unit = tiles of a full unit
  unsigned ntu_magic;  // ceil(2^32 / ntu) (unused when ntu == 1)
  int claim_limit;     // ntu * (total_units + 4): no CU can legitimately claim more tiles
  int mode0, mode1;    // requant mode of stage 0 / stage 1: 0 exact, 1 fast, 2 magic (see header)
  int s0_uniform;      // fused op with ONE conv0 scale (count 1, op_conv.cc:311-313): s0_value, no per-channel reads
  float s0_value;
  int tile_stride;     // bytes between the MFMA_NB input-tile slots in LDS
  int static_rounds;   // units a loader owns statically before it turns to the queue
  int pool;            // unfused ops only: 1 = 2x2 stride-2 max pooling fused into the store stage.  Units are
                       // full-width row groups (th even); a tile is 2 rows x 16 columns, so that a pooling
                       // window is accumulator registers e, e+1, e+8, e+9 of one lane; dst has oh/2 x ow/2 pixels
  int lazy_queue;      // 1: a loader draws its next unit only when the slot for it is free (store-bound ops)
  int half_from;       // unit ids >= half_from denote HALF units (th / 2 rows): id half_from + 2 i + j is half j of unit
                       // half_from + i; total_units counts them.  INT_MAX: none.  The tail of a store-bound op: the
                       // queue's granule is what a workgroup still holds when the queue runs dry (dfx_api.hip)
  int *queue;          // [0] next unit, [1] finished loaders; both 0 between launches
#ifdef DFX_STAMPS
  unsigned long long *prof;  // diagnostic build only: [workgroup][wave][16] cycle sums
#endif
#ifdef DFX_TRACE
  int *trace;  // diagnostic build only (make trace): host-visible [workgroup][wave][4] progress words
#endif
};

#ifdef DFX_TRACE
// progress word of this wave in host-pinned memory: readable by the host while the kernel runs
#define DFX_TRACE_AT(code, v0, v1)                                                                      \
  do {                                                                                                  \
    if ((int)threadIdx.x == __builtin_amdgcn_readfirstlane((int)threadIdx.x)) { /* first ACTIVE lane */ \
      int *tr_ = g.trace + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 4;                          \
      __hip_atomic_fetch_add(tr_ + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); /* events so far */ \
      if ((threadIdx.x & 63) != 0 && tr_[1] == 0) /* sticky: first trace point lane 0 was missing at */   \
        __hip_atomic_store(tr_ + 1, (int)(code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);           \
      (void)(v0);                                                                                       \
      __hip_atomic_store(tr_ + 2, (int)__builtin_amdgcn_read_exec_lo(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); (void)(v1); \
      __hip_atomic_store(tr_ + 0, (int)(code), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);            \
    }                                                                                                   \
  } while (0)
#else
#define DFX_TRACE_AT(code, v0, v1)
#endif

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
// cycle sums live in LDS (behind the control block), not in registers: the kernel sits at the 128-VGPR /
// ~100-SGPR limit, and register-resident sums made the stamps build spill inside the store loop.
// All 64 lanes add (the read-out divides by 64).
#define DFX_ACC(slot, expr) prof_acc[slot] += (expr)  // conv_stream.cuh / conv_direct.cuh: register sums
// timeline of this wave (stamps build): event n = {v0, v1, v2, v3}; all lanes store the same words
#define DFX_TLOG(n, v0, v1, v2, v3)                                                                      \
  do {                                                                                                   \
    if ((n) < 8) {                                                                                       \
      unsigned long long *tl_ = g.prof + (size_t)gridDim.x * 256 +                                       \
                                (((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (n)) * 4;          \
      tl_[0] = (v0); tl_[1] = (v1); tl_[2] = (v2); tl_[3] = (v3);                                        \
    }                                                                                                    \
  } while (0)
#define DFX_LACC(slot, expr)                                                                             \
  __hip_atomic_fetch_add(ctrl + 32 + 8 * (int)(threadIdx.x >> 6) + (slot), (int)(unsigned)(expr), __ATOMIC_RELAXED, \
                         __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define DFX_STAMP(var)
#define DFX_ACC(slot, expr)
#define DFX_LACC(slot, expr)
#define DFX_TLOG(n, v0, v1, v2, v3)
#endif

// f32 value of an accumulator before scaling: vcvtdq2ps(acc) + bias, with the
// u8->s8 compensation folded in as an exact f32 add (conv_stream.cuh / conv_direct.cuh)
__device__ __forceinline__ float acc_to_f32(int raw, float comp, float bias) {
  return __fadd_rn(__fadd_rn(__int2float_rn(raw), comp), bias);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// DFX_PACKED_F32 (never defined in the product build): round 2 wrote the requant arithmetic with v_pk_add_f32 /
// v_pk_mul_f32 to halve its instruction count.  On gfx950 the packed-f32 instructions execute on the MATRIX
// pipe's side of the SIMD: they do not overlap with MFMAs -- in one wave's stream every one of them adds ~10
// cycles to an MFMA-paced loop, and issued by another wave of the SIMD they run at 13.8 cycles each beside an
// MFMA stream -- whereas plain v_add_f32 / v_mul_f32 / v_fma_f32 / v_cvt_pk_u8_f32 hide under it: six per MFMA
// in the same wave cost nothing, another wave's run at 6.5 cycles each (tools/probe/probe_coexec.hip,
// profiles/r03/probe_coexec.jsonl).  All requant arithmetic is therefore written with scalar float operations
// and the library is built with -fno-slp-vectorize so that hipcc does not pack them again;
// tests/test_isa_hazards_cpu.py checks that no packed-f32 instruction is left in the conv kernels.

#ifndef DFX_RING
#define DFX_RING 5  // conv0 fragment prefetch depth (k-steps in flight): LDS latency is several hundred cycles under load
#endif

// The output is written once and never re-read by this kernel: non-temporal stores
// keep it from displacing the input rows / weights in L2 and leave fewer dirty lines
// to write back at the end of the kernel.
#ifndef DFX_TEMPORAL_STORES
#define DFX_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))  // <= 8 bytes
#define DFX_STORE16(ptr, val) dfx_store16_nt((ptr), (val))             // 16 bytes: see dfx_device.cuh
#else
#define DFX_STORE(ptr, val) (*(ptr) = (val))
#define DFX_STORE16(ptr, val) dfx_store16((ptr), (val))
#endif

// ---- conv_stream.cuh / conv_direct.cuh: one pixel's G consecutive channels -> one store.
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
  if (FAST) {  // bs holds comp + bias.  Plain v_add_f32 / v_mul_f32, never the packed forms: see DFX_PACKED_F32
#pragma unroll
    for (int c = 0; c < G; ++c) f[c] = __fmul_rn(__fadd_rn(__int2float_rn(acc[c]), bs[c]), sc[c]);
  } else {
#pragma unroll
    for (int c = 0; c < G; ++c) f[c] = __fmul_rn(acc_to_f32(acc[c], cp[c], bs[c]), sc[c]);
  }
  if (DST == DFX_F32) {
#pragma unroll
    for (int c = 0; c < G; ++c) f[c] = relu ? relu_x86(f[c]) : f[c];
    if (G == 4) DFX_STORE16(reinterpret_cast<v4f *>(p), (v4f{f[0], f[1], f[2], f[3]}));
    else if (G == 2) *reinterpret_cast<float2 *>(p) = float2{f[0], f[1]};
    else *reinterpret_cast<float *>(p) = f[0];
  } else if (DST == DFX_S32) {
    int v[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      if (FAST) v[c] = (int)__builtin_rintf(relu ? __builtin_fmaxf(f[c], 0.0f) : f[c]);
      else v[c] = cvt_x86_rt(relu ? relu_x86(f[c]) : f[c], rm);
    }
    if (G == 4) DFX_STORE16(reinterpret_cast<v4i *>(p), (v4i{v[0], v[1], v[2], v[3]}));
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

// ---- this kernel's store stage.  f[] = one pixel's G consecutive channels after scaling -> ReLU,
// conversion, one typed store ----
// sat_u8_bits for the exact-mode code of the resident kernel: the literal is opaque, or hipcc hoists
// 0xff, 0xff00, ... into VGPRs that stay live across the whole (register-bound) tile loop
__device__ __forceinline__ unsigned sat_u8_bits_cold(int v) {
  unsigned lim = 255u;
  asm volatile("" : "+v"(lim));
  return min((unsigned)v, lim);
}

template <int DST, int G, bool FAST>
__device__ __forceinline__ void store_pixel(unsigned char *p, float (&f)[G], bool relu, int rm) {
  if (DST == DFX_F32) {
#pragma unroll
    for (int c = 0; c < G; ++c) f[c] = relu ? relu_x86(f[c]) : f[c];
    if (G == 4) DFX_STORE16(reinterpret_cast<v4f *>(p), (v4f{f[0], f[1], f[2], f[3]}));
    else if (G == 2) *reinterpret_cast<float2 *>(p) = float2{f[0], f[1]};
    else *reinterpret_cast<float *>(p) = f[0];
  } else if (DST == DFX_S32) {
    int v[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      if (FAST) v[c] = (int)__builtin_rintf(relu ? __builtin_fmaxf(f[c], 0.0f) : f[c]);
      else v[c] = cvt_x86_rt(relu ? relu_x86(f[c]) : f[c], rm);
    }
    if (G == 4) DFX_STORE16(reinterpret_cast<v4i *>(p), (v4i{v[0], v[1], v[2], v[3]}));
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
        const unsigned b = (DST == DFX_U8) ? sat_u8_bits_cold(v) : ((unsigned)sat_s8(v) & 0xffu);
        pk |= b << (8 * c);
      }
    }
    if (G == 4) DFX_STORE(reinterpret_cast<unsigned *>(p), pk);
    else if (G == 2) *reinterpret_cast<unsigned short *>(p) = (unsigned short)pk;
    else *p = (uint8_t)pk;
  }
}

// TWO pixels (accumulator registers e and e + 1, e even: adjacent registers, so the packed f32
// instructions take them as they stand) x G consecutive channels -> two stores.
// `acc` are accumulator BITS that started from MAGIC1_BITS; per-lane constants ia, fb, fc:
//   MODE 2 (magic)  f = (as_float(acc) + fb) * fc        fb = (comp + bias - 2^23 - 0x22F983) * 2^-26,
//                                                        fc = scale * 2^26
//   MODE 1 (fast)   f = (float(acc + ia) + fb) * fc      ia = comp - MAGIC1_BITS, fb = bias, fc = scale
//   MODE 0 (exact)  the same with the x86 conversions and selects
template <int DST, int G, int MODE>
__device__ __forceinline__ void emit_pair(unsigned char *p0, unsigned char *p1, const v16i (&acc)[G], int e,
                                          const int (&ia)[G], const v2f (&fb)[G], const v2f (&fc)[G],
                                          bool relu, int rm, bool w0 = true, bool w1 = true) {
  float f0[G], f1[G];
#pragma unroll
  for (int c = 0; c < G; ++c) {
    if (MODE == 3) {  // fb = the exact product (comp + bias - m / ulp) * scale, fc = scale / ulp: one rounding
      f0[c] = __builtin_fmaf(__int_as_float(acc[c][e]), fc[c][0], fb[c][0]);
      f1[c] = __builtin_fmaf(__int_as_float(acc[c][e + 1]), fc[c][0], fb[c][0]);
    } else if (MODE == 2) {
#ifndef DFX_PACKED_F32  // plain v_add_f32 + v_mul_f32 (see DFX_PACKED_F32 above)
      f0[c] = __fmul_rn(__fadd_rn(__int_as_float(acc[c][e]), fb[c][0]), fc[c][0]);
      f1[c] = __fmul_rn(__fadd_rn(__int_as_float(acc[c][e + 1]), fb[c][0]), fc[c][0]);
#else
      v2f x = {__int_as_float(acc[c][e]), __int_as_float(acc[c][e + 1])};
      x = (x + fb[c]) * fc[c];  // fb[c] = {k, k}, fc[c] = {s, s}: plain 64-bit operands, no op_sel
      f0[c] = x[0];
      f1[c] = x[1];
#endif
    } else {
      f0[c] = __fmul_rn(__fadd_rn(__int2float_rn(acc[c][e] + ia[c]), fb[c][0]), fc[c][0]);
      f1[c] = __fmul_rn(__fadd_rn(__int2float_rn(acc[c][e + 1] + ia[c]), fb[c][0]), fc[c][0]);
    }
  }
  if (w0) store_pixel<DST, G, MODE != 0>(p0, f0, relu, rm);  // (w0 / w1: partial tiles, pixels beyond the tile's end)
  if (w1) store_pixel<DST, G, MODE != 0>(p1, f1, relu, rm);
}

// One POOLED pixel: maximum over accumulator registers e, e+1 (two columns of the window's first row) and
// e+8, e+9 (second row), G consecutive channels -> one store.  The reference's planned op is conv -> relu ->
// pooling (test/test_conv_relu_pooling.cc:30-235), i.e. the maximum of the four CONVERTED pixels:
//   integer dst, MODE != 0: relu, nearest-even rounding and saturation are monotonic, so the maximum is taken
//     on the scaled f32 values and converted once;
//   f32 dst: relu per pixel, then the pooling kernel's chain acc = acc > x ? acc : x in window order (signed
//     zeros and the order of equal values as in pool_eltwise.hip);
//   MODE 0 (x86 overflow / NaN results are not monotonic): every pixel is converted, the maximum is taken on
//     the converted values.
template <int DST, int G, int MODE>
__device__ __forceinline__ void emit_pool(unsigned char *p, const v16i (&acc)[G], int e, const int (&ia)[G],
                                          const v2f (&fb)[G], const v2f (&fc)[G], bool relu, int rm, bool w) {
  float f[4][G];
#pragma unroll
  for (int c = 0; c < G; ++c)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = e + 8 * q;
      if (MODE == 2) {
        f[2 * q][c] = __fmul_rn(__fadd_rn(__int_as_float(acc[c][r]), fb[c][0]), fc[c][0]);
        f[2 * q + 1][c] = __fmul_rn(__fadd_rn(__int_as_float(acc[c][r + 1]), fb[c][0]), fc[c][0]);
      } else {
        f[2 * q][c] = __fmul_rn(__fadd_rn(__int2float_rn(acc[c][r] + ia[c]), fb[c][0]), fc[c][0]);
        f[2 * q + 1][c] = __fmul_rn(__fadd_rn(__int2float_rn(acc[c][r + 1] + ia[c]), fb[c][0]), fc[c][0]);
      }
    }
  if (DST == DFX_F32) {
    float m[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      float accm = -__builtin_inff();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = relu ? relu_x86(f[q][c]) : f[q][c];
        accm = accm > x ? accm : x;
      }
      m[c] = accm;
    }
    if (w) {
      if (G == 4) DFX_STORE16(reinterpret_cast<v4f *>(p), (v4f{m[0], m[1], m[2], m[3]}));
      else if (G == 2) *reinterpret_cast<float2 *>(p) = float2{m[0], m[1]};
      else *reinterpret_cast<float *>(p) = m[0];
    }
  } else if (MODE != 0) {
    float m[G];
#pragma unroll
    for (int c = 0; c < G; ++c) m[c] = __builtin_fmaxf(__builtin_fmaxf(f[0][c], f[1][c]), __builtin_fmaxf(f[2][c], f[3][c]));
    if (w) store_pixel<DST, G, true>(p, m, relu, rm);
  } else {
    int v[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      int best = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = cvt_x86_rt(relu ? relu_x86(f[q][c]) : f[q][c], rm);
        const int t = DST == DFX_U8 ? (int)sat_u8_bits_cold(x) : DST == DFX_S8 ? sat_s8(x) : x;
        best = q == 0 ? t : max(best, t);
      }
      v[c] = best;
    }
    if (w) {
      if (DST == DFX_S32) {
        if (G == 4) DFX_STORE16(reinterpret_cast<v4i *>(p), (v4i{v[0], v[1], v[2], v[3]}));
        else if (G == 2) *reinterpret_cast<int2 *>(p) = int2{v[0], v[1]};
        else *reinterpret_cast<int *>(p) = v[0];
      } else {
        unsigned pk = 0;
#pragma unroll
        for (int c = 0; c < G; ++c) pk |= ((unsigned)v[c] & 0xffu) << (8 * c);
        if (G == 4) DFX_STORE(reinterpret_cast<unsigned *>(p), pk);
        else if (G == 2) *reinterpret_cast<unsigned short *>(p) = (unsigned short)pk;
        else *p = (uint8_t)pk;
      }
    }
  }
}

// First MFMA of a chain whose accumulator starts from the inline constant 1/(2*pi) (MAGIC1_BITS).
// Written as asm because hipcc, given the constant as a v16i splat used by several MFMAs,
// materialises it in 16 VGPRs per use instead of the inline operand.  hipcc does not look into
// asm blocks: the wait states it would put between a VALU instruction that writes A / B and the
// MFMA (it uses 2; that read-after-write dependency is not interlocked: configuration B_valu_raw of
// tools/probe/probe_mfma_war.hip fails without them) are in the block.
__device__ __forceinline__ v16i mfma_i8_from_magic(v4i a, v4i b) {
  v16i d;
  asm volatile("s_nop 3\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 0.15915494" : "=&v"(d) : "v"(a), "v"(b));
  return d;
}

// FUSED = false is the unfused conv() overload (reference deepfusion.h:121-129): the same
// loader / tile machinery and 3x3 MFMA pipeline, but the contraction is oriented
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
  // LDS: [control block | W0 fragments | W1 fragments | constants | MFMA_NB input tiles].  The control
  // block sits at LDS address 0: ds_append takes its address from M0 + a 16-bit immediate, and only
  // the form M0 = 0 is relied upon here.
  int *ctrl = reinterpret_cast<int *>(smem);
  unsigned char *w0s = smem + MFMA_CTRL_BYTES;                 // [OCB][9][ICB][64 lanes][16 B]
  unsigned char *w1s = w0s + OCB * 9 * ICB * 1024;             // [NCB][OCB][64 lanes][16 B]
  float *cst = reinterpret_cast<float *>(w1s + NCB * OCB * 1024);
  const int cst_bytes = (mfma_cst_floats(OC, OC1) * 4 + 15) & ~15;
  unsigned char *tiles = reinterpret_cast<unsigned char *>(cst) + cst_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform
  const int team = wave >> 3, cw = wave & 7;                  // cw == MFMA_CW: loader `team`
  const int LW = g.tw + 2;
  const int upi = g.uy * g.ux;

  // ---- halo-tile chunk helpers (loader waves; compute waves for the very first tiles) ----
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
  // The halo tile in LDS is an array of 16-byte GRANULES in row-major (tile row, tile column, slot)
  // order; slot `sl` of the pixel in tile column X holds the pixel's channel chunk sl ^ swz(X).
  // Granule q of a unit, general form: branch-free with CLAMPED, always-in-range coordinates --
  // granules outside the image read a valid pixel of the same image and are replaced by zeros
  // (real 0 = the stored byte 0x80).  q must be < tile_chunks.
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
  // unit -> (image, unit row, unit column) with multiply-high instead of integer division (a unit id is
  // < 2^31 / 64, far inside the range where ceil(2^32 / d) is exact)
  // (hrow, hth: first row inside the unit and rows of the piece the id denotes -- the whole unit, or one half)
  auto unit_split = [&](int unit, int &n, int &uyi, int &uxi, int &hrow, int &hth) {
    hrow = 0;
    hth = g.th;
    if (unit >= g.half_from) {
      const int v = unit - g.half_from;
      unit = g.half_from + (v >> 1);
      hth = g.th >> 1;
      hrow = (v & 1) * hth;
    }
    n = g.upi_magic ? (int)__umulhi((unsigned)unit, g.upi_magic) : unit;
    const int u = unit - n * upi;
    uyi = g.ux_magic ? (int)__umulhi((unsigned)u, g.ux_magic) : u;
    uxi = u - uyi * g.ux;
  };
  // (a half unit's halo tile is loaded like a whole unit's, th + 2 rows from its own first row: one row too many)
  auto unit_origin = [&](int unit, const uint8_t *&src_n, int &y0, int &x0) {
    int n, uyi, uxi, hrow, hth;
    unit_split(unit, n, uyi, uxi, hrow, hth);
    y0 = uyi * g.th + hrow - a.pt;
    x0 = uxi * g.tw - a.pl;
    src_n = a.src + (size_t)n * a.ih * a.iw * IC;
  };

  // what a compute wave needs to know about a unit (the loader computes it once per unit)
  auto unit_info = [&](int unit, int &pix0, int &thtw, int &tprm) {
    int n, uyi, uxi, hrow, hth;
    unit_split(unit, n, uyi, uxi, hrow, hth);
    const int y0 = uyi * g.th + hrow, x0 = uxi * g.tw;
    const int th = max(0, min(hth, a.oh - y0)), tw = min(g.tw, a.ow - x0);  // (0 rows: the second half of a bottom unit)
    const int tpr = g.pool ? (tw + 15) >> 4 : (tw + 31) >> 5;
    pix0 = g.pool ? (n * (a.oh >> 1) + (y0 >> 1)) * (a.ow >> 1) + (x0 >> 1)  // first POOLED pixel of the unit
                  : (n * a.oh + y0) * a.ow + x0;
    thtw = (th << 16) | tw;
    tprm = tpr > 1 ? (int)(((1ull << 32) + tpr - 1) / tpr) : 0;
  };

  // The FIRST tile of each loader's stream (k = 0, 1) is staged cooperatively by 7 compute
  // waves each right after the weights (2-3 chunks per thread, one memory round trip), and
  // published before the barrier: the loader's own start-up (per-lane staging table,
  // 22-chunk prefetch address math) used to sit between the barrier and the first MFMA.
  // The loaders then start with k = 2, 3.  Needs statically known first units.
  // Stream id of loader `tm` of this workgroup, team-major: a partial last static round (stream ids
  // below total_units % T) then gives every workgroup one more unit instead of two more to half of them.
  // Workgroup id with XCD-major numbering (round 3): hardware deals workgroups round-robin over the 8 XCDs, so
  // the units of consecutive ids -- vertical neighbours that share two halo rows -- landed on eight different L2s
  // and every halo row was fetched from HBM twice.  Renumbered (blockIdx % 8 picks the group: speed only),
  // neighbours run on one XCD in the same round and the second reader hits in L2.
  const int wg = (gridDim.x % 8 == 0) ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8)
                                      : (int)blockIdx.x;
  auto stream_id = [&](int tm) { return tm * (int)gridDim.x + wg; };
  const bool coop0 = g.static_rounds >= 1;
  // (Rounds 1-3 staged every stream's SECOND unit before the barrier as well -- static_rounds >= 2 -- so that all
  // four slots were full when the claim loop started.  But then the whole chip pulls four units per CU from HBM at
  // once and every workgroup waits for all of it; with the loader fetching its second unit behind the barrier like
  // every later one the barrier is passed ~3.7 k cycles earlier and the headline launches are 0.6-0.8 us shorter:
  // profiles/r03/ab_s32_second_unit.txt, and conv_mfma_roles.cuh's loader for what else was tried.)

  // LDS control words are read and written by WHOLE waves (every lane the same word, the same
  // value) and every loaded value goes through readfirstlane, so that all control flow below is
  // scalar: hipcc (ROCm 7.2) otherwise treats the per-lane results of the atomic loads as
  // divergent, wraps the claim loop in exec-mask bookkeeping and -- with an `if (lane == 0)`
  // around the claiming atomic -- left lane 0 masked off after the first tile (the wave then
  // re-claimed tile 0 forever; found with the progress-trace build, profiles/debug/trace_hang.py).
  auto ctl_load = [&](int idx) {
    return __builtin_amdgcn_readfirstlane(
        __hip_atomic_load(ctrl + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
  };
  auto ctl_store = [&](int idx, int v) {
    __hip_atomic_store(ctrl + idx, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // ---- weights + constants: the host keeps them in ONE device buffer laid out
  //      exactly like the LDS image [W0 fragments | W1 fragments | constants], so a
  //      single linear copy stages them.  All global loads of a pass are issued
  //      before the first LDS write: one memory round trip per 128 KB. ----
  auto stage_weights = [&]() {  // called by the 14 compute waves (the loaders hold tile data)
    constexpr int NT = MFMA_TEAMS * MFMA_CW * 64;
    constexpr int TT = MFMA_CW * 64;  // threads of the 7 compute waves that stage one first tile
    const int ctid = (team * MFMA_CW + cw) * 64 + lane, tctid = cw * 64 + lane;
    const v4i *s = reinterpret_cast<const v4i *>(a.wei);
    v4i *d = reinterpret_cast<v4i *>(w0s);
    const int total = OCB * 9 * ICB * 64 + NCB * OCB * 64 + (mfma_cst_floats(OC, OC1) * 4 + 15) / 16;
    // first tile of stream `team` (coop0): ring slot `team`, unit blockIdx * 2 + team; its first
    // 4 chunks per thread travel together with the weights (one memory round trip for both)
    unsigned char *slot = tiles + (size_t)team * g.tile_stride;
    const int unit0 = stream_id(team);
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
        u0 = load_granule(src_n, y0, x0, min(tctid + 0 * TT, g.tile_chunks - 1));
        u1 = load_granule(src_n, y0, x0, min(tctid + 1 * TT, g.tile_chunks - 1));
        u2 = load_granule(src_n, y0, x0, min(tctid + 2 * TT, g.tile_chunks - 1));
        u3 = load_granule(src_n, y0, x0, min(tctid + 3 * TT, g.tile_chunks - 1));
      }
      d[min(q0 + 0 * NT, last)] = t0;
      d[min(q0 + 1 * NT, last)] = t1;
      d[min(q0 + 2 * NT, last)] = t2;
      d[min(q0 + 3 * NT, last)] = t3;
      if (tile0 && base == 0) {
        if (tctid + 0 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (tctid + 0 * TT)) = u0;
        if (tctid + 1 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (tctid + 1 * TT)) = u1;
        if (tctid + 2 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (tctid + 2 * TT)) = u2;
        if (tctid + 3 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (tctid + 3 * TT)) = u3;
      }
    }
    if (tile0) {  // a tile of more than 4 chunks per thread: the rest
      for (int base = 4 * TT; base < g.tile_chunks; base += 4 * TT) {
        const int q0 = base + tctid;
        const v4i t0 = load_granule(src_n, y0, x0, min(q0 + 0 * TT, g.tile_chunks - 1));
        const v4i t1 = load_granule(src_n, y0, x0, min(q0 + 1 * TT, g.tile_chunks - 1));
        const v4i t2 = load_granule(src_n, y0, x0, min(q0 + 2 * TT, g.tile_chunks - 1));
        const v4i t3 = load_granule(src_n, y0, x0, min(q0 + 3 * TT, g.tile_chunks - 1));
        if (q0 + 0 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (q0 + 0 * TT)) = t0;
        if (q0 + 1 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (q0 + 1 * TT)) = t1;
        if (q0 + 2 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (q0 + 2 * TT)) = t2;
        if (q0 + 3 * TT < g.tile_chunks) *reinterpret_cast<v4i *>(slot + 16 * (q0 + 3 * TT)) = t3;
      }
    }
    if (ctid < MFMA_CTRL_BYTES / 4) {
      // control block: zero; "no end seen yet"; with cooperatively staged first tiles, slots 0
      // and 1 start out published (or their stream is marked empty)
      int v = 0;
      if (ctid == CTL_END || ctid == CTL_END + 1) {
        v = 0x7fffffff;
        if (coop0 && stream_id(ctid - CTL_END) >= g.total_units) v = ctid - CTL_END;
      }
      // slot sl starts out published when its unit is staged before the barrier: slots 0, 1 by the
      // compute waves (coop0); slots 2, 3 are published by the loaders behind the barrier
      auto first_unit = [&](int sl) {
        const int u = (sl >> 1) * (int)gridDim.x * MFMA_TEAMS + stream_id(sl & 1);
        return sl < 2 && coop0 && u < g.total_units ? u : -1;
      };
      if (ctid >= CTL_FULL && ctid < CTL_FULL + MFMA_NB && first_unit(ctid - CTL_FULL) >= 0) v = 1;
      if (ctid >= CTL_INFO && ctid < CTL_INFO + 4 * MFMA_NB) {
        const int u0 = first_unit((ctid - CTL_INFO) >> 2);
        if (u0 >= 0) {
          int i0, i1, i2;
          unit_info(u0, i0, i1, i2);
          const int f = (ctid - CTL_INFO) & 3;
          v = f == 0 ? i0 : f == 1 ? i1 : f == 2 ? i2 : 0;
        }
      }
      ctrl[ctid] = v;
    }
  };

  if (cw == MFMA_CW) {
    // =========================== loader wave `team` ===========================
    // owns k = team, team + 2, ...: ring slot k % 4 = 2 * (j & 1) + team for its j-th unit.
    // Highest issue priority: the loaders execute ~300 instructions per unit against the compute
    // waves' thousands, but as the youngest waves of their SIMDs they lost every arbitration and
    // needed ~10 k cycles per unit for them (profiles/stamps.py), which left the 14 compute waves
    // waiting for tiles 40 % of the time.
#ifndef DFX_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    v4i pf[MFMA_LC];
    // Per-lane tables for granule lane + 64 i (fixed for the whole launch):
    //   rel[i]   byte offset of its SOURCE chunk from the unit's first halo pixel (tile row 0, column 0)
    // FAST units -- every tile row inside the image with a row to spare above and below, full width:
    // the 22 loads are then base + rel[i] with a scalar base and no further arithmetic; the (at most two)
    // padding columns read the neighbouring row's pixels instead of zeros and are overwritten after
    // the tile has been written (padfix).  Other units (image borders) take the general, clamped form.
    // (kept as 16-bit granule offsets, two per register: with the 88 data registers a full 32-bit
    // table spills to scratch, and every reload then waits for all loads in flight)
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
    // padfix entries e = lane, lane + 64: (side, tile row, slot) -> LDS byte offset of that granule
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
    // issue-early half: first 64*MFMA_LC granules of a unit -> registers
#define DFX_PREFETCH(UNIT)                                                              \
  do {                                                                                  \
    const uint8_t *base_u_;                                                             \
    cur_fast = unit_fast((UNIT), base_u_, cur_left, cur_right) ? 1 : 0;                 \
    if (cur_fast) {                                                                     \
      /* unconditional (rel is clamped into the tile): a load inside a branch makes hipcc wait vmcnt(0) there */ \
      _Pragma("unroll") for (int i = 0; i < MFMA_LC; ++i) {                             \
        unsigned rp_ = relp[i / 2]; /* opaque: unpacked here, not hoisted (and spilled) */ \
        asm volatile("" : "+v"(rp_));                                                   \
        pf[i] = *reinterpret_cast<const v4i *>(base_u_ + ((i % 2 ? rp_ >> 16 : rp_ & 0xffffu) << 4)); \
      }                                                                                 \
    } else {                                                                            \
      const uint8_t *src_n_;                                                            \
      int y0_, x0_;                                                                     \
      unit_origin((UNIT), src_n_, y0_, x0_);                                            \
      int lq_ = lane; /* opaque: keep the per-granule index math out of the LICM set */ \
      asm volatile("" : "+v"(lq_));                                                     \
      _Pragma("unroll") for (int i = 0; i < MFMA_LC; ++i)                               \
          pf[i] = load_granule(src_n_, y0_, x0_, min(lq_ + 64 * i, g.tile_chunks - 1)) ^ x80; \
    }                                                                                   \
  } while (0)
    // (the general form leaves pf as u8 with zero padding, like the fast form: the write applies ^ 0x80)
    // Unit sequence of this loader: the first `static_rounds` units are owned
    // statically (round j -> unit j*T + stream id; no atomic: 2 x gridDim loaders
    // hammering one queue word at kernel start cost ~12 us), the rest come from the
    // device-side queue.  Draws are device-scope atomics that take microseconds to
    // return: one is kept in flight and only broadcast (readfirstlane = wait) when needed.
    const int T = (int)gridDim.x * MFMA_TEAMS, tg = stream_id(team);
    const bool use_queue = g.static_rounds * T < g.total_units;  // false: the static rounds cover the op
    auto unit_at = [&](int j) {
      int v = j * T + tg;
      if (j >= g.static_rounds) {
        v = 0x7fffffff;
        if (use_queue && lane == 0) v = g.static_rounds * T + atomicAdd(g.queue, 1);
      }
      return v;
    };

    // coop0: the compute waves stage this stream's first tile themselves; the loader passes
    // the barrier at once (nobody waits for its start-up) and begins with its second unit
    auto write_tile = [&](unsigned char *ins) {
      {  // write-late half of the staging: granule lane + 64 i -> LDS byte 16 * (lane + 64 i)
        // (branch-free: pieces beyond the tile go to the 1 KB dump piece behind it)
        unsigned char *dst = ins + lane * 16;
        const int dump = g.tile_stride - 1024;
#pragma unroll
        for (int i = 0; i < MFMA_LC; ++i)
          *reinterpret_cast<v4i *>(dst + (64 * i < g.tile_chunks ? 1024 * i : dump)) = pf[i] ^ x80;
      }
      if (cur_fast) {  // padding columns of a fast unit (after the tile writes: same wave, LDS keeps the order)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          if ((padside[m] == 0 && cur_left) || (padside[m] == 1 && cur_right))
            *reinterpret_cast<v4i *>(ins + padoff[m]) = x80;
      }
    };
    auto write_rest = [&](unsigned char *ins, int unit) {  // oversized tile: the rest is staged synchronously
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
    DFX_STAMP(l_pre);
    const int j0 = coop0 ? 1 : 0;
    if (coop0) __syncthreads();
    // lazy: store-bound ops, whose workgroups run at very different speeds (their share of the HBM write
    // bandwidth: lifetimes 100 k .. 206 k cycles on the s32 headline).  Drawing ahead -- one unit per
    // loader is staged before the barrier, the next fetched behind it, one more prefetched, one more drawn -- hands out all 3.5
    // units per loader in the first quarter of the kernel, so the queue balances nothing.  A lazy loader
    // draws when the slot for the unit is free: fast workgroups come back for more.  The ~6 us from
    // draw to published tile are covered by the three other slots (a tile takes ~16 us here).
    const bool lazy = use_queue && g.lazy_queue;
    int cur = lazy ? 0 : __builtin_amdgcn_readfirstlane(unit_at(j0));
    int nxt_v = lazy ? 0 : unit_at(j0 + 1);
    int jn = j0 + 2;
    if (!lazy && cur < g.total_units) DFX_PREFETCH(cur);  // (!coop0: the first tile's loads fly during the weight copy)
    if (!coop0) __syncthreads();  // the only workgroup barrier: weights + control block are in LDS
    DFX_STAMP(l_post);
#ifdef DFX_STAMPS
    if (lane == 0) {  // loader: cycles from entry to its first loop iteration
      unsigned long long *o = g.prof + ((size_t)blockIdx.x * 16 + wave) * 16;
      o[13] = l_pre - t_entry;
      o[14] = l_post - l_pre;
    }
#endif

    DFX_TRACE_AT(100, cur, j0);
    for (int j = j0;; ++j) {
      DFX_STAMP(la);
      const int s = 2 * (j & 1) + team, gen = j >> 1;
      DFX_TRACE_AT(101, cur, j);
      unsigned char *ins = tiles + (size_t)s * g.tile_stride;
      if (lazy) {  // wait for the slot first, draw (a synchronous device-scope atomic) only then
        for (int spin = 0; spin < MFMA_SPIN_LIMIT && ctl_load(CTL_DONE + s) < 64 * g.ntu * gen; ++spin)
          __builtin_amdgcn_s_sleep(8);
        cur = __builtin_amdgcn_readfirstlane(unit_at(j));
        if (cur < g.total_units) DFX_PREFETCH(cur);
      }
      const bool valid = cur < g.total_units;
      if (!valid) {  // this stream has ended at k = 2j + team (a stream that coop0 found empty is marked already)
        if (!(coop0 && j == 1 && tg >= g.total_units)) ctl_store(CTL_END + team, 2 * j + team);
        break;
      }
      // slot s is free once every tile claim of its previous generation was counted off
      for (int spin = 0; spin < MFMA_SPIN_LIMIT && ctl_load(CTL_DONE + s) < 64 * g.ntu * gen; ++spin)
        __builtin_amdgcn_s_sleep(2);
      DFX_STAMP(lb);
      write_tile(ins);
      write_rest(ins, cur);
      DFX_TRACE_AT(102, cur, j);
      {
        int i0, i1, i2;
        unit_info(cur, i0, i1, i2);
        ctrl[CTL_INFO + 4 * s + 0] = i0;
        ctrl[CTL_INFO + 4 * s + 1] = i1;
        ctrl[CTL_INFO + 4 * s + 2] = i2;
      }
      // publish: LDS executes a wave's DS instructions in order; the release makes the
      // compiler keep them ahead of the flag store and waits for them to complete
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // all lanes' tile writes first
      ctl_store(CTL_FULL + s, gen + 1);
      DFX_STAMP(lc);
      if (!lazy) {
        cur = __builtin_amdgcn_readfirstlane(nxt_v);
        nxt_v = unit_at(jn++);
        if (cur < g.total_units) DFX_PREFETCH(cur);
      }
      DFX_STAMP(ld);
#ifdef DFX_STAMPS
      // slot wait, tile write + publish, next draw + prefetch issue, units
      DFX_LACC(0, lb - la); DFX_LACC(1, lc - lb); DFX_LACC(2, ld - lc); DFX_LACC(3, 1);
      DFX_TLOG(j - j0, la - t_entry, lb - t_entry, lc - t_entry, ((unsigned long long)(ld - t_entry) << 16) | (unsigned)(2 * j + team));
#endif
    }
#ifdef DFX_STAMPS
    if (lane == 0) {
      unsigned long long *o = g.prof + ((size_t)blockIdx.x * 16 + wave) * 16;
      for (int k = 0; k < 4; ++k) o[k] = (unsigned)ctrl[32 + 8 * wave + k] / 64u;
    }
#endif
#undef DFX_PREFETCH
    // last loader out re-arms the queue for the next launch.  The draw still in
    // flight must have been performed before this loader counts itself out, or it
    // could land after the reset.
    DFX_TRACE_AT(103, cur, 0);
    const int pending = __builtin_amdgcn_readfirstlane(nxt_v);
    if (use_queue && lane == 0 && pending >= 0) {
      const int fin = atomicAdd(g.queue + 1, 1);
      if (fin == (int)gridDim.x * MFMA_TEAMS - 1) {
        atomicExch(g.queue, 0);
        atomicExch(g.queue + 1, 0);
      }
    }
    DFX_TRACE_AT(199, 0, 0);
    return;
  }

  // =========================== compute waves ===========================
  DFX_TRACE_AT(1, 0, 0);
  stage_weights();
  DFX_TRACE_AT(2, 0, 0);
  DFX_STAMP(t_staged);  // (diagnostic builds: the stamp waits for this wave's LDS writes)
  __syncthreads();
  // constants in LDS: [A0 | B0 | C0] per conv0 channel, [A1 | B1 | C1] per 1x1 channel
  const int *ia0 = reinterpret_cast<const int *>(cst);
  const float *fb0 = cst + OC, *fc0 = cst + 2 * OC;                    // fused: scalars per conv0 channel
  const v2f *pb0 = reinterpret_cast<const v2f *>(cst + OC), *pc0 = reinterpret_cast<const v2f *>(cst + 3 * OC);  // unfused: pairs
  const int *ia1 = reinterpret_cast<const int *>(cst + 3 * OC);
  const v2f *pb1 = reinterpret_cast<const v2f *>(cst + 3 * OC + OC1), *pc1 = reinterpret_cast<const v2f *>(cst + 3 * OC + 3 * OC1);
  const bool relu1 = (FUSED ? a.relu1 : a.relu0) || DST == DFX_U8;  // ReLU of the stage that stores
  const int mode0 = g.mode0, mode1 = g.mode1;
  const unsigned row_bytes = (unsigned)(FUSED ? OC1 : OC) * ESZ;  // dst bytes per pixel
  const int lds_row = LW * IC;                                    // bytes per halo-tile row in LDS

  DFX_STAMP(t_loop);
#ifdef DFX_STAMPS
  int tl_n = 0;
#endif
  DFX_LACC(4, t_loop - t_entry);  // start-up: weights staging + barrier
#ifdef DFX_STAMPS
  const unsigned long long startup_stage = t_staged - t_entry;
#endif
  DFX_TRACE_AT(3, 0, 0);
  // Tile claims.  ds_append returns the counter's old value to the whole wave (wave-uniform) and adds
  // the number of active lanes, 64: claim = old value >> 6.  It is an ordinary LDS instruction with a
  // returned value, so the wait for it sits where the value is first used.  LDS round trips take
  // ~1 k cycles while 14 waves stream fragments, therefore the control traffic of tile i + 1 is issued
  // inside tile i (see "claim-ahead distance" below).  A claim is always processed by the wave that drew
  // it (a wave that leaves holds one beyond the end).
  typedef __attribute__((address_space(3))) int lds_int;
  auto draw = [&]() { return __builtin_amdgcn_ds_append((lds_int *)(ctrl + CTL_NEXT)); };
  struct Look { int full, end; v4i info; };
  // one batch of LDS reads: FULL first, then the slot's unit record (LDS serves a wave's reads in
  // order and the loader wrote the record before FULL), then the stream's END
  auto look = [&](int sl, int par) {
    Look l;
    l.full = __hip_atomic_load(ctrl + CTL_FULL + sl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    // (three LDS word loads: a volatile vector load through a generic pointer becomes flat_load + s_waitcnt
    // vmcnt(0), i.e. a wait for every outstanding output store)
    l.info[0] = __hip_atomic_load(ctrl + CTL_INFO + 4 * sl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    l.info[1] = __hip_atomic_load(ctrl + CTL_INFO + 4 * sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    l.info[2] = __hip_atomic_load(ctrl + CTL_INFO + 4 * sl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    l.info[3] = 0;
    l.end = __hip_atomic_load(ctrl + CTL_END + par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return l;
  };
  auto split = [&](int t, int &k, int &ti) {
    k = g.ntu == 1 ? t : (int)__umulhi((unsigned)t, g.ntu_magic);  // (ceil(2^32 / 1) does not fit 32 bits)
    ti = t - k * g.ntu;
  };
  // Claim-ahead distance: the NEXT tile's claim is drawn after this tile's conv0 and its flags / unit
  // record are looked at in the last store group (unfused: claim before conv0, look after it) -- late
  // enough that a claimed tile does not sit unprocessed for a whole tile time (with only 28 tiles in the
  // ring, 14 parked claims delayed every slot's release by a round and starved the third round), early
  // enough to hide the ~1 k-cycle LDS round trips.  The first claim of a wave is synchronous.
  int c_ahead = 0;
  int t = __builtin_amdgcn_readfirstlane(draw()) >> 6, k, ti;
  split(t, k, ti);
  Look lk = look(k & (MFMA_NB - 1), k & 1);
  for (;;) {
    DFX_STAMP(c0);
    DFX_TRACE_AT(4, t, g.claim_limit);
    if (t > g.claim_limit) break;  // cannot happen (a CU never claims more than every tile of the op): keeps a logic error from hanging the GPU
    const int s = k & (MFMA_NB - 1), gen = k >> 2, p = k & 1;
    bool have = false;
    v4i info = lk.info;
    for (int spin = 0; spin < MFMA_SPIN_LIMIT; ++spin) {  // (bounded: a protocol error must not hang the GPU)
      if (__builtin_amdgcn_readfirstlane(lk.full) >= gen + 1) { have = true; info = lk.info; break; }
      if (__builtin_amdgcn_readfirstlane(lk.end) <= k) break;
      __builtin_amdgcn_s_sleep(1);
      lk = look(s, p);
    }
    const int t_cur = t, k_cur = k, ti_cur = ti;
    (void)t_cur;
    bool drawn = false, looked = false;
    auto next_draw = [&]() { c_ahead = draw(); drawn = true; };
    auto next_look = [&]() {
      t = __builtin_amdgcn_readfirstlane(c_ahead) >> 6;
      split(t, k, ti);
      lk = look(k & (MFMA_NB - 1), k & 1);
      looked = true;
    };
    DFX_TRACE_AT(5, t_cur, have);
    DFX_STAMP(c1);
    DFX_LACC(0, c1 - c0);  // wait for the tile's unit (claims and looks are prefetched)
    if (!have) {  // stream p has no k-th unit; done when the other stream has none for k + 1 either
#ifdef DFX_STAMPS
      DFX_TLOG(tl_n, c0 - t_entry, c1 - t_entry, c1 - t_entry, ((unsigned long long)t_cur << 8));
      ++tl_n;
#endif
      if (ctl_load(CTL_END + (p ^ 1)) <= k_cur + 1) break;
      next_draw();
      next_look();
      continue;
    }
    const int ti_now = ti_cur;
    const unsigned char *ins = tiles + (size_t)s * g.tile_stride;
    const int pix0 = __builtin_amdgcn_readfirstlane(info[0]);
    const int thtw = __builtin_amdgcn_readfirstlane(info[1]);
    const int th = thtw >> 16, tw = thtw & 0xffff;
    const int npx = th * tw;
    const int tiles_per_row = (tw + 31) >> 5;
    const int tiles_per_row16 = (tw + 15) >> 4;  // pooled ops: tiles of 2 rows x 16 columns
    const int ntiles = g.pool ? (th >> 1) * tiles_per_row16 : g.linear ? (npx + 31) >> 5 : th * tiles_per_row;

    DFX_TRACE_AT(6, pix0, ntiles);
    if (ti_now < ntiles) {
      // Lane-derived values are recomputed per tile from the lane id (two v_mbcnt, volatile so that it
      // is not hoisted): kept in registers across the loop they are what hipcc spills, and a
      // scratch reload waits (vmcnt) for every output store of the previous tile.
      int lane_t;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
      const int lane = lane_t, l31 = lane_t & 31, h = lane_t >> 5;
      // pixel of this lane (conv0 column) and the tile's output base (wave-uniform)
      int ty, tx, nvalid;
      size_t obase;  // dst pixel index of px_local == 0
      if (g.pool) {  // nvalid = valid COLUMNS of the 16-column block
        const int tprm = __builtin_amdgcn_readfirstlane(info[2]);
        const int tr = tprm ? (int)__umulhi((unsigned)ti_now, (unsigned)tprm) : ti_now, tc = ti_now - tr * tiles_per_row16;
        nvalid = min(16, tw - 16 * tc);
        ty = 2 * tr + (l31 >> 4);
        tx = 16 * tc + min(l31 & 15, nvalid - 1);
        obase = (size_t)pix0 + (size_t)tr * (a.ow >> 1) + 8 * tc;  // pooled pixel index
      } else if (g.linear) {
        nvalid = min(32, npx - 32 * ti_now);
        const int pc = 32 * ti_now + min(l31, nvalid - 1);
        ty = tw == 1 ? pc : (int)__umulhi((unsigned)pc, g.tw_magic);  // tw == g.tw in linear mode
        tx = pc - ty * tw;
        obase = (size_t)pix0 + 32 * ti_now;
      } else {
        const int tprm = __builtin_amdgcn_readfirstlane(info[2]);
        const int tr = tprm ? (int)__umulhi((unsigned)ti_now, (unsigned)tprm) : ti_now, tc = ti_now - tr * tiles_per_row;
        nvalid = min(32, tw - 32 * tc);
        ty = tr;
        tx = 32 * tc + min(l31, nvalid - 1);
        obase = (size_t)pix0 + (size_t)tr * a.ow + 32 * tc;
      }
      nvalid = __builtin_amdgcn_readfirstlane(nvalid);  // wave-uniform: keep it in a scalar register
      // Lane-constant LDS offsets are made opaque once per tile: otherwise LICM
      // hoists every weight / constant fragment read out of the tile loop and
      // keeps >200 VGPRs live across it (spills).
      int lane16 = lane * 16, h4 = 4 * h, lch = G * l31;
      asm volatile("" : "+v"(lane16), "+v"(h4), "+v"(lch));

      DFX_TRACE_AT(60, ty, tx);
      DFX_STAMP(c2);
      // ---- conv0: 9 taps x ICB k-steps x OCB row blocks ----
      // per-lane B-fragment addresses for tap row 0: one per (tap column, ic half); tap rows 1, 2
      // are lds_row bytes further each.  The chunk swizzle depends on the LDS column only.
      int bb[3][ICB];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int X = tx + dx;
        const int pb = (ty * LW + X) * IC, sw = chunk_swizzle<CP>(X);
#pragma unroll
        for (int c = 0; c < ICB; ++c) bb[dx][c] = pb + 16 * ((2 * c + h) ^ sw);
      }
      if (!FUSED) next_draw();
      v16i acc0[OCB];
      if (FUSED) {  // start values per channel from LDS (0 / comp / bits(1.5 * 2^23) + comp + bias)
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4i iv = *reinterpret_cast<const v4i *>(ia0 + 32 * r + 8 * q + h4);
            acc0[r][4 * q + 0] = iv[0]; acc0[r][4 * q + 1] = iv[1];
            acc0[r][4 * q + 2] = iv[2]; acc0[r][4 * q + 3] = iv[3];
          }
      }  // (unfused: the first MFMA of each chain starts from the inline constant MAGIC1_BITS)
      {
        // software pipeline: the fragments of k-step s+RD-1 are fetched right after the
        // MFMAs of step s were issued (bounded prefetch depth; sched_barrier keeps hipcc
        // from hoisting all 54 loads to the top and spilling)
        constexpr int NS = 9 * ICB;  // k-steps
        constexpr int RD = DFX_RING;
        v4i fb[RD], fw[RD][OCB];
        auto fetch = [&](int st, int slot) {  // st, slot are compile-time after unrolling
          const int tap = st / ICB, c = st % ICB;
          fb[slot] = *reinterpret_cast<const v4i *>(ins + (tap / 3) * lds_row + bb[tap % 3][c]);
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
          for (int r = 0; r < OCB; ++r) {
            if (FUSED) acc0[r] = mfma_i8(fw[st % RD][r], fb[st % RD], acc0[r]);        // D0[oc][px]
            else if (st == 0) acc0[r] = mfma_i8_from_magic(fb[st % RD], fw[st % RD][r]);
            else acc0[r] = mfma_i8(fb[st % RD], fw[st % RD][r], acc0[r]);              // D0[px][oc]
          }
          __builtin_amdgcn_sched_barrier(0);
          if (st + RD - 1 < NS) fetch(st + RD - 1, (st + RD - 1) % RD);
        }
      }

      if (FUSED) next_draw(); else next_look();
      DFX_TRACE_AT(61, 0, 0);
      unsigned char *tile_dst = reinterpret_cast<unsigned char *>(a.dst) + obase * row_bytes;
      using T = std::true_type;
      using F = std::false_type;
      if constexpr (!FUSED) {
        // ---- unfused: requant 0 + typed store straight from the 3x3 accumulators ----
        const int chb = lch;  // G == OCB: this lane owns channels G*l31 .. G*l31 + G-1
        int ia[G];
        v2f fb[G], fc[G];
#pragma unroll
        for (int cc = 0; cc < G; ++cc) {
          ia[cc] = ia0[chb + cc];
          fb[cc] = pb0[chb + cc];
          fc[cc] = pc0[chb + cc];
        }
        // Partial tiles (check_tag): the lanes beyond nvalid loaded the tile's LAST valid pixel
        // (min(l31, nvalid - 1) above); the stores of pixels beyond the tile's end are predicated off
        // (addresses stay clamped in range).  Round 2's first version stored those copies to the last
        // pixel's address instead: 12.5 % more store traffic with 2-row units (PMC WRITE_SIZE).
        // All addresses are the wave-uniform tile_dst plus a 32-bit per-lane byte offset.
        const unsigned lane_off = (unsigned)h4 * row_bytes + (unsigned)chb * ESZ;
        const unsigned ch_off = (unsigned)chb * ESZ;
        auto emit0 = [&](auto mode_tag, auto check_tag) {
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const int pl = 8 * (e >> 2) + (e & 3);  // + 4h: pixel of register e; e + 1 is the next pixel
            unsigned char *p0, *p1;  // (full tiles: scalar pixel base + one per-lane offset, see the fused stage)
            bool w0 = true, w1 = true;
            if (decltype(check_tag)::value) {
              w0 = pl + h4 < nvalid;
              w1 = pl + 1 + h4 < nvalid;
              p0 = tile_dst + ((unsigned)min(pl + h4, nvalid - 1) * row_bytes + ch_off);
              p1 = tile_dst + ((unsigned)min(pl + 1 + h4, nvalid - 1) * row_bytes + ch_off);
            } else {
              p0 = (tile_dst + (size_t)((unsigned)pl * row_bytes)) + (size_t)lane_off;
              p1 = (tile_dst + (size_t)((unsigned)(pl + 1) * row_bytes)) + (size_t)lane_off;
            }
            emit_pair<DST, G, decltype(mode_tag)::value>(p0, p1, acc0, e, ia, fb, fc, relu1, a.rm0, w0, w1);
          }
        };
        // fused 2x2/2 max pooling: pooled column 4*((e>>2)&1) + 2h + ((e&3)>>1) of the tile's 8, from registers
        // e, e+1, e+8, e+9 (e = 0, 2, 4, 6); a pooled pixel exists when both of its columns do
        auto emit0p = [&](auto mode_tag) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const int pcl = 4 * ((e >> 2) & 1) + ((e & 3) >> 1) + (h4 >> 1);
            const bool w = 2 * pcl + 1 < nvalid;
            unsigned char *p = tile_dst + ((unsigned)min(pcl, max(0, (nvalid >> 1) - 1)) * row_bytes + ch_off);
            emit_pool<DST, G, decltype(mode_tag)::value>(p, acc0, e, ia, fb, fc, relu1, a.rm0, w);
          }
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        if (g.pool) {
          if (mode0 == 2) emit0p(M2{}); else if (mode0 == 1) emit0p(M1{}); else emit0p(M0{});
        }
        else if (mode0 == 2) { if (nvalid == 32) emit0(M2{}, F{}); else emit0(M2{}, T{}); }
        else if (mode0 == 1) { if (nvalid == 32) emit0(M1{}, F{}); else emit0(M1{}, T{}); }
        else { if (nvalid == 32) emit0(M0{}, F{}); else emit0(M0{}, T{}); }
      } else {
      DFX_STAMP(c3);
      DFX_LACC(1, c3 - c2);  // conv0 MFMA issue
      // ---- requant 0 in registers -> A fragments of the 1x1 ----
      v4i mid[OCB];
      if (mode0 == 3) {  // "fma": bits = 2^23 + t (t >= 0) or 2^23 - |t| / 2: one v_fma_f32(x, s, -2^23 s), see conv_mfma_roles.cuh
        const bool uni = g.s0_uniform != 0;
        const float su = g.s0_value, cu = -8388608.0f * g.s0_value;
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v4f sc = {su, su, su, su}, cc = {cu, cu, cu, cu};
            if (!uni) {
              sc = *reinterpret_cast<const v4f *>(fc0 + 32 * r + 8 * q + h4);
              cc = *reinterpret_cast<const v4f *>(fb0 + 32 * r + 8 * q + h4);
            }
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(__int_as_float(acc0[r][4 * q + i]), sc[i], cc[i]), i, pk);
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      } else if (mode0 == 2) {  // accumulator bits are the float 1.5 * 2^23 + acc + bias: subtract, scale, pack
        const bool uni = g.s0_uniform != 0;  // (the op's single scale: no per-channel reads)
        const float su = g.s0_value;
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v4f sc = {su, su, su, su};
            if (!uni) sc = *reinterpret_cast<const v4f *>(fc0 + 32 * r + 8 * q + h4);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__fmul_rn(__fadd_rn(__int_as_float(acc0[r][4 * q + i]), -MAGIC0_F), sc[i]), i, pk);
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      } else if (mode0 == 1) {  // ReLU + RNE + saturation + pack in one op
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ch = 32 * r + 8 * q + h4;
            const v4f bs = *reinterpret_cast<const v4f *>(fb0 + ch);
            const v4f sc = *reinterpret_cast<const v4f *>(fc0 + ch);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              pk = __builtin_amdgcn_cvt_pk_u8_f32(__fmul_rn(__fadd_rn(__int2float_rn(acc0[r][4 * q + i]), bs[i]), sc[i]), i, pk);
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      } else {
#pragma unroll
        for (int r = 0; r < OCB; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ch = 32 * r + 8 * q + h4;
            const v4f bs = *reinterpret_cast<const v4f *>(fb0 + ch);
            const v4f sc = *reinterpret_cast<const v4f *>(fc0 + ch);
            unsigned pk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float f = __fmul_rn(__fadd_rn(__int2float_rn(acc0[r][4 * q + i]), bs[i]), sc[i]);
              pk |= sat_u8_bits_cold(cvt_x86_rt(relu_x86(f), a.rm0)) << (8 * i);
            }
            mid[r][q] = (int)(pk ^ 0x80808080u);
          }
      }
      DFX_TRACE_AT(62, mode0, mode1);
      DFX_STAMP(c4);
      DFX_LACC(2, c4 - c3);  // requant 0
      // The claim drawn after conv0 has arrived: move it to a scalar register, the store loop below
      // needs every VGPR (a spill there makes each tile wait for all of its stores).
      c_ahead = __builtin_amdgcn_readfirstlane(c_ahead);
#ifndef DFX_NO_SETPRIO
      {
        // Issue priority by how far this wave lags: the SIMD arbiter prefers its oldest wave, so the
        // youngest of four needed 3x as long per tile (stamps timeline: 12 k vs 33..43 k cycles) -- and a
        // slot is only refilled when the LAST tile of its unit is done, so the stragglers stalled
        // everyone.  The distance between this tile's claim and the one just drawn is ~14 when the
        // wave keeps pace with the other 13.
        const int lag = (c_ahead >> 6) - t_cur;
        if (lag > 24) __builtin_amdgcn_s_setprio(3);
        else if (lag > 18) __builtin_amdgcn_s_setprio(2);
        else if (lag > 12) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
      }
#endif
      // ---- conv1 + requant 1 + store, G column blocks at a time ----
      for (int cg = 0; cg < NCG; ++cg) {
        if (cg == NCG - 1) next_look();
        DFX_TRACE_AT(63, cg, NCG);
        // (per group from the lane id, itself two v_mbcnt: no live VGPR across the groups instead of three)
        int lt;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lt));
        const int lane16 = lt * 16, lch = G * (lt & 31), h4 = 4 * (lt >> 5);
        const int chb = 32 * G * cg + lch;  // this lane's first channel in the group
        v16i acc1[G];
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
          for (int cc = 0; cc < G; ++cc)
            acc1[cc] = r == 0 ? mfma_i8_from_magic(mid[0], wf[0][cc]) : mfma_i8(mid[r], wf[r][cc], acc1[cc]);
        // OCB == 1: every MFMA above is the asm form, whose result hipcc does not know to be an MFMA
        // result: it would not pad the 12 wait states a VALU read of it needs (probe_mfma_raw).
        if (OCB == 1) asm volatile("s_nop 7\n\ts_nop 4" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // requant constants of this lane's channels: fetched behind the MFMAs (their LDS latency
        // overlaps the matrix pipe; loaded earlier they sit on top of accumulators + weight
        // fragments, the kernel's register peak)
        int ia[G];
        v2f fb[G], fc[G];
#pragma unroll
        for (int cc = 0; cc < G; ++cc) {
          ia[cc] = ia1[chb + cc];
          fb[cc] = pb1[chb + cc];
          fc[cc] = pc1[chb + cc];
        }
        DFX_TRACE_AT(64, cg, nvalid);
        // register e of the accumulator = pixel 8*(e>>2) + 4h + (e&3) of the tile
        // (opaque per group: keeps hipcc from hoisting 16 per-pixel offsets -- of both variants -- out of
        // the group loop and spilling them to scratch)
        unsigned rb = row_bytes;
        int nv1 = nvalid - 1;
        asm volatile("" : "+s"(rb), "+s"(nv1));
        const int h4c = h4;
        const unsigned lane_off = (unsigned)h4c * rb + (unsigned)chb * ESZ;
        const unsigned ch_off = (unsigned)chb * ESZ;
        auto emit = [&](auto mode_tag, auto check_tag) {  // (partial tiles: see the unfused stage above)
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const int pl = 8 * (e >> 2) + (e & 3);  // + 4h: pixel of register e; e + 1 is the next pixel
            // full tiles: scalar pixel base (SALU) + ONE per-lane offset for all 16 stores -- the VALU is
            // this kernel's bound (4 cycles per wave instruction per SIMD), per-pixel address arithmetic in
            // vector registers cost a quarter of the epilogue's instructions
            unsigned char *p0, *p1;
            bool w0 = true, w1 = true;
            if (decltype(check_tag)::value) {  // partial tile: pixels beyond its end are not written
              w0 = pl + h4c <= nv1;            // (uniform per half wave: the pixel depends on the register and h only)
              w1 = pl + 1 + h4c <= nv1;
              p0 = tile_dst + ((unsigned)min(pl + h4c, nv1) * rb + ch_off);
              p1 = tile_dst + ((unsigned)min(pl + 1 + h4c, nv1) * rb + ch_off);
            } else {
              p0 = (tile_dst + (size_t)((unsigned)pl * rb)) + (size_t)lane_off;
              p1 = (tile_dst + (size_t)((unsigned)(pl + 1) * rb)) + (size_t)lane_off;
            }
            emit_pair<DST, G, decltype(mode_tag)::value>(p0, p1, acc1, e, ia, fb, fc, relu1, a.rm1, w0, w1);
          }
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        if (mode1 == 2) { if (nvalid == 32) emit(M2{}, F{}); else emit(M2{}, T{}); }
        else if (mode1 == 1) { if (nvalid == 32) emit(M1{}, F{}); else emit(M1{}, T{}); }
        else { if (nvalid == 32) emit(M0{}, F{}); else emit(M0{}, T{}); }
        DFX_TRACE_AT(66, cg, 0);
      }
      DFX_STAMP(c5);
      DFX_LACC(3, c5 - c4);  // conv1 + requant 1 + stores
      }  // FUSED
      DFX_LACC(6, 1);
    }
    if (!drawn) next_draw();  // (a claim beyond its unit's tiles: nothing was computed)
    if (!looked) next_look();
    DFX_TRACE_AT(7, t_cur, 0);
    DFX_STAMP(c6);
#ifdef DFX_STAMPS
    DFX_TLOG(tl_n, c0 - t_entry, c1 - t_entry, c6 - t_entry, ((unsigned long long)t_cur << 8) | (have ? 1 : 0));
    ++tl_n;
#endif
    // count this claim off on its slot (the tile's LDS reads have been consumed by the MFMAs)
    __hip_atomic_fetch_add(ctrl + CTL_DONE + s, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // all 64 lanes: + 64
    DFX_LACC(5, c6 - c0);  // whole claim
    DFX_LACC(7, 1);
  }
  DFX_TRACE_AT(9, 0, 0);
#ifdef DFX_STAMPS
  {
    DFX_STAMP(t_end);
    unsigned long long rt;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
    if (lane == 0) {
      unsigned long long *o = g.prof + ((size_t)blockIdx.x * 16 + wave) * 16;
      for (int k = 0; k < 8; ++k) o[k] = (unsigned)ctrl[32 + 8 * wave + k] / 64u;
      o[8] = t_entry; o[9] = t_end; o[10] = rt; o[11] = rt_entry; o[12] = startup_stage;
    }
  }
#endif
}

}  // namespace dfx
