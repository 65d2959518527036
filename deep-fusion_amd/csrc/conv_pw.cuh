// conv_pw.cuh -- unfused POINTWISE conv (1x1 window, stride 1, no padding) u8 x s8 -> s32 -> requant, as one
// int8-MFMA GEMM per 32 output pixels whose pixel fragments go straight from global memory into the MFMA operand
// registers (gfx950 / CDNA4).  No input tile in LDS at all.
//
// Replaces, for this shape class (ResNet "reduce" convs: 256 -> 64 at 56 x 56, 512 -> 128 at 28 x 28):
//   compute_loop / store_output   /root/reference/src/jit_conv_kernel.cc:317-393, :218-305 (kh = kw = 1)
//   infer_conv0                   src/op_conv.cc:31-138
// Same arithmetic and requant routes as conv_direct.cuh's unfused mode (bit-exact against the oracle on the same
// cases); what differs is the data path.  A pointwise conv is HBM-bound by its input (pw256: 103 MB in, 26 MB out,
// 13 GOP): conv_stream.cuh, which stages a halo tile and streams weights through LDS per 128 pixels, reaches 3.0 TB/s
// on it (42 us); conv_direct.cuh 55 us (profiles/r03/unfused_pointwise.txt).  Here
//  * the weights (oc * ic <= 96 KB, packed as conv_direct.cuh's W0d with one tap) and the constants sit in LDS for
//    the whole launch (LDS-DMA at entry, one barrier);
//  * a wave owns 32-pixel blocks b = wave id, wave id + #waves, ...; lane (pixel p, k half h) reads its 16 bytes of
//    every 32-channel k-block with one global_load_dwordx4 at an IMMEDIATE offset from its pixel's row pointer
//    (a pixel's two lanes cover 32 contiguous bytes: whole sectors), xors them to s8 and feeds them to the MFMA as
//    the B operand; loads run a ring of two 4-k-block chunks (8 KB per wave) ahead, across block ends, branch-free;
//  * the epilogue assembles the block's output rows in a wave-private LDS area (a lane holds 4 consecutive channels of
//    its pixel per quarter block) and stores 16 bytes per lane: whole contiguous rows.  (Stored straight from the
//    accumulators the 8-byte pieces were not merged by the L2: 4.2 x write traffic, 80 us.)
// Supported: kh = kw = 1, stride 1, padding 0, ic a multiple of 256, oc in {64, 128, 256}, oc * ic <= 96 KB; all dst
// types and requant routes.  Everything else pointwise stays on conv_stream.cuh.
#pragma once

#include "conv_mfma.cuh"

namespace dfx {

constexpr int PW_THREADS = 256;
constexpr int PW_CH = 4;  // k-blocks per chunk (a lane's 4 x 16 bytes at offsets 0, 32, 64, 96 of a 128-byte line)

struct PwGeom {
  int icb, ocb;      // 32-channel blocks
  int px_total;      // bs * oh * ow (< 2^31 / ic: byte offsets of pixel rows stay in 32 bits per lane pair ... see host)
  int n_blocks;      // ceil(px_total / 32)
  int off_cst;       // LDS byte offset of the constants (behind the weights)
  int off_stage;     // ... of the waves' store staging (behind the constants), stage_bytes per wave
  int stage_bytes;
  int fast, m0;      // host proofs: fast requant route valid; stage-0 "fma" route (u8 dst)
};

// one quarter block (4 consecutive channels of the lane's pixel) after requant: the packed dword (1-byte outputs) or the
// four 4-byte values' bit patterns.  The arithmetic is store_group's / pack_group's (conv_mfma.cuh), value for value.
template <int DST, bool FAST>
__device__ __forceinline__ v4i pw_quarter(const int (&acc4)[4], const v4i cp4, const v4f bs4, const v4f sc4, bool relu, int rm, bool fma0) {
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  v4i out = {0, 0, 0, 0};
  if (DST == DFX_U8 && fma0) {
    unsigned pk = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(__int_as_float(acc4[i]), sc4[i], bs4[i]), i, pk);
    out[0] = (int)pk;
    return out;
  }
  int v[4];
  float bsa[4], sca[4], zf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = acc4[i] + (FAST ? 0 : cp4[i]);
    bsa[i] = bs4[i]; sca[i] = sc4[i]; zf[i] = 0.0f;
  }
  if constexpr (ESZ == 1) {
    out[0] = (int)pack_group<DST, 4, FAST>(v, zf, bsa, sca, relu, rm);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float f = FAST ? __fmul_rn(__fadd_rn(__int2float_rn(v[i]), bsa[i]), sca[i]) : __fmul_rn(acc_to_f32(v[i], zf[i], bsa[i]), sca[i]);
      if (DST == DFX_F32) out[i] = __float_as_int(relu ? relu_x86(f) : f);
      else out[i] = FAST ? (int)__builtin_rintf(relu ? __builtin_fmaxf(f, 0.0f) : f) : cvt_x86_rt(relu ? relu_x86(f) : f, rm);
    }
  }
  return out;
}

template <int OCB, int DST>
__global__ __launch_bounds__(PW_THREADS, 2) void conv_pw_kernel(ConvArgs a, PwGeom g) {
  constexpr int ESZ = (DST == DFX_F32 || DST == DFX_S32) ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const w_lds = smem;                                   // W0d[ob][kb][lane][16]
  const float *const cst = reinterpret_cast<const float *>(smem + g.off_cst);
  const int OCP = 32 * g.ocb;
  const int *comp0 = reinterpret_cast<const int *>(cst);
  const float *bias0 = cst + OCP, *scale0 = cst + 2 * OCP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  {  // weights + constants -> LDS by LDS-DMA (1 KB per wave instruction); [W0d | consts] is contiguous in global memory
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void global_void;
    const int wq = g.ocb * g.icb * 64;                  // 16-byte chunks of weights
    const int total16 = wq + 3 * OCP / 4;
    const v4i *ws = reinterpret_cast<const v4i *>(a.wei);
    const v4i *cs = reinterpret_cast<const v4i *>(a.consts);
    v4i *wd = reinterpret_cast<v4i *>(smem);
    for (int j = wave; 64 * j < total16; j += PW_THREADS / 64) {
      const int q = 64 * j + lane;
      // (the constants start on a 1 KB boundary of the LDS image as well: wq is a multiple of 64)
      if (q < wq) __builtin_amdgcn_global_load_lds((global_void *)(ws + q), (lds_void *)(wd + 64 * j), 16, 0, 0);
      else if (q < total16) __builtin_amdgcn_global_load_lds((global_void *)(cs + (q - wq)), (lds_void *)(wd + 64 * j), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int gw = blockIdx.x * (PW_THREADS / 64) + wave, GW = gridDim.x * (PW_THREADS / 64);
  const int nch = g.icb / PW_CH;  // chunks per block (even: ic is a multiple of 256)
  const bool relu0 = a.relu0 != 0 || DST == DFX_U8;
  const bool fast = g.fast != 0, fma0 = DST == DFX_U8 && g.m0 != 0;
  const v4i x80 = v4i{(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
  const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned char *const dst_b = reinterpret_cast<unsigned char *>(a.dst);
  const unsigned row_bytes = (unsigned)a.oc * ESZ;
  // this lane's slice of the weight image: ob's fragments start at wa[ob]
  int wa[OCB];
#pragma unroll
  for (int ob = 0; ob < OCB; ++ob) wa[ob] = lane * 16 + ob * g.icb * 1024;

  auto row_ptr = [&](int b) -> const unsigned char * {  // lane's 16 bytes of k-block 0 of pixel 32 b + l31 (clamped to the last pixel)
    const int px = min(32 * b + l31, g.px_total - 1);
    return a.src + (size_t)px * (size_t)a.ic + 16 * h;
  };
  v4i fx[2][PW_CH];
  auto fetch = [&](int set, const unsigned char *p) {  // one chunk: immediate offsets 0, 32, 64, 96
#pragma unroll
    for (int j = 0; j < PW_CH; ++j) fx[set][j] = *reinterpret_cast<const v4i *>(p + 32 * j);
  };
  if (gw >= g.n_blocks) return;
  const unsigned char *xp = row_ptr(gw);
  fetch(0, xp);
  fetch(1, xp + 32 * PW_CH);
  for (int b = gw; b < g.n_blocks; b += GW) {
    // (past the wave's last block: a harmless re-read of ITS OWN block -- not of the batch's last block: 5 k waves
    // re-reading the same 8 KB at the end of the launch serialised on one L2 channel, 80 us instead of 25)
    const unsigned char *xn = row_ptr(b + GW < g.n_blocks ? b + GW : b);
    v16i acc[OCB];
    if (fma0) {  // "fma": start from bits(2^23) + comp + bias of this lane's 16 channels (comp slot of the constants)
#pragma unroll
      for (int ob = 0; ob < OCB; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4i iv = *reinterpret_cast<const v4i *>(comp0 + ob * 32 + 8 * q + 4 * h);
          acc[ob][4 * q + 0] = iv[0]; acc[ob][4 * q + 1] = iv[1]; acc[ob][4 * q + 2] = iv[2]; acc[ob][4 * q + 3] = iv[3];
        }
    } else {
#pragma unroll
      for (int ob = 0; ob < OCB; ++ob) acc[ob] = zero16;
    }
    for (int c = 0; c < nch; c += 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int cc = c + s;  // this chunk; its ring set is s (nch is even)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < PW_CH; ++j) {
          const v4i bfrag = fx[s][j] ^ x80;  // u8 -> s8 (the exact compensation 128 * sum(w) is in comp0)
#pragma unroll
          for (int ob = 0; ob < OCB; ++ob) {
            const v4i wfrag = *reinterpret_cast<const v4i *>(w_lds + wa[ob] + (cc * PW_CH + j) * 1024);
            acc[ob] = mfma_i8(wfrag, bfrag, acc[ob]);  // D[oc][px]
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // refill this set with the chunk two ahead: of this block, or the first two of the wave's next block
        const int ca = cc + 2;
        fetch(s, ca < nch ? xp + 32 * PW_CH * ca : xn + 32 * PW_CH * (ca - nch));
      }
    }
    // ---- requant + store.  Lane = pixel 32 b + l31; per output block and quarter q it holds channels 8 q + 4 h .. + 3.
    // Stored straight from there a wave instruction would write 32 scattered 8-byte (or 32-byte) pieces; the L2 does
    // not merge them into lines (PMC: 108 MB written for a 26 MB output, 80 us).  So the block's rows are assembled
    // in a wave-private LDS area first and leave as 16 bytes per lane: 1-byte outputs whole pixel rows (the block's
    // 32 rows are ONE contiguous 32 * oc bytes of dst), 4-byte outputs one output block (128 bytes per pixel) at a time.
    unsigned char *stg = smem + g.off_stage + wave * g.stage_bytes;
    const int nvalid = min(32, g.px_total - 32 * b);
    unsigned char *dst_blk = dst_b + (size_t)(32 * b) * row_bytes;
    auto quarter = [&](int ob, int q) -> v4i {
      const int ch = ob * 32 + 8 * q + 4 * h;
      const v4f bs4 = *reinterpret_cast<const v4f *>(bias0 + ch);
      const v4f sc4 = *reinterpret_cast<const v4f *>(scale0 + ch);
      v4i cp4 = {0, 0, 0, 0};
      if (!fast) cp4 = *reinterpret_cast<const v4i *>(comp0 + ch);
      int a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = acc[ob][4 * q + i];
      return fast ? pw_quarter<DST, true>(a4, cp4, bs4, sc4, relu0, a.rm0, fma0) : pw_quarter<DST, false>(a4, cp4, bs4, sc4, relu0, a.rm0, fma0);
    };
    if constexpr (ESZ == 1) {
      const int pitch = a.oc + 16;  // (row pitch of the staging: odd multiple of 16 for oc = 64 / 128 / 256)
#pragma unroll
      for (int ob = 0; ob < OCB; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<int *>(stg + l31 * pitch + ob * 32 + 8 * q + 4 * h) = quarter(ob, q)[0];
      const int c16n = a.oc >> 4;  // 16-byte chunks per pixel row: 4, 8 or 16 (a power of two)
      const int sh = c16n == 4 ? 2 : c16n == 8 ? 3 : 4;
      for (int ck = lane; ck < 32 * c16n; ck += 64) {
        const int row = ck >> sh, c16 = ck & (c16n - 1);
        const v4i val = *reinterpret_cast<const v4i *>(stg + row * pitch + 16 * c16);
        if (row < nvalid) DFX_STORE16(reinterpret_cast<v4i *>(dst_blk + (size_t)row * row_bytes + 16 * c16), val);
      }
    } else {
#pragma unroll
      for (int ob = 0; ob < OCB; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v4i *>(stg + l31 * 144 + 32 * q + 16 * h) = quarter(ob, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // 32 rows x 128 bytes = 256 chunks
          const int ck = lane + 64 * k, row = ck >> 3, c16 = ck & 7;
          const v4i val = *reinterpret_cast<const v4i *>(stg + row * 144 + 16 * c16);
          if (row < nvalid) DFX_STORE16(reinterpret_cast<v4i *>(dst_blk + (size_t)row * row_bytes + ob * 128 + 16 * c16), val);
        }
      }
    }
    xp = xn;
  }
}

}  // namespace dfx
