// dfx_api.hip -- host side of the C ABI declared in include/dfx.h.
//
// Host-side counterparts of the reference's op drivers:
//   validation        op_conv<T>::init_conf + jit_conv_kernel::init_conf
//                     (/root/reference/src/op_conv.cc:262-365, src/jit_conv_kernel.cc:512-673)
//   workspace/buffers op_conv<T>::op_conv (src/op_conv.h:70-95), util/memory.cc:21-40
//   dispatch          op_conv<T>::infer (src/op_conv.cc:22-29) -> ONE kernel launch
//   concat            op_concat<T> (src/op_concat.h:28-61), jit_concat_kernel::init_conf
//                     (src/jit_concat_kernel.cc:130-197)
// No CPU compute path exists here: without a usable HIP device every compute
// entry point returns an error.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "conv_mfma.cuh"
#include "conv_stream.cuh"
#include "conv_direct.cuh"
#include "conv_pw.cuh"
#include "conv_mfma_roles.cuh"
#include "dfx_device.cuh"

namespace dfx {
int launch_conv_generic(const ConvArgs &a, hipStream_t s, int *grid_out, int *lds_out);
#define DFX_DECL(n) \
  int launch_conv_mfma_##n(const ConvArgs &, const MfmaGeom &, int, int, int, int, int, hipStream_t, int)
DFX_DECL(f32);
DFX_DECL(s32);
DFX_DECL(s8);
DFX_DECL(u8);
#undef DFX_DECL
int launch_conv_mfma_roles_u8(const ConvArgs &, const MfmaGeom &, int, int, int, int, int, hipStream_t, int);
int launch_conv_mfma_roles_s8(const ConvArgs &, const MfmaGeom &, int, int, int, int, int, hipStream_t, int);
#define DFX_DECL(n) \
  int launch_conv_mfma_##n##_unfused(const ConvArgs &, const MfmaGeom &, int, int, int, int, hipStream_t, int)
DFX_DECL(f32);
DFX_DECL(s32);
DFX_DECL(s8);
DFX_DECL(u8);
#undef DFX_DECL

#define DFX_DECL(n) \
  int launch_conv_stream_##n(const ConvArgs &, const StreamGeom &, int, int, int, int, int, int, hipStream_t, int)
DFX_DECL(f32);
DFX_DECL(s32);
DFX_DECL(s8);
DFX_DECL(u8);
#undef DFX_DECL

#define DFX_DECL(n) \
  int launch_conv_direct_##n(const ConvArgs &, const DirectGeom &, int, int, int, int, int, int, hipStream_t, int)
DFX_DECL(f32);
DFX_DECL(s32);
DFX_DECL(s8);
DFX_DECL(u8);
#undef DFX_DECL

int launch_concat(const ConcatArgs &a, hipStream_t s);
int launch_pool(const PoolArgs &a, hipStream_t s);
int launch_eltwise(const EltwiseArgs &a, hipStream_t s);
}  // namespace dfx

using namespace dfx;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(e_ == hipErrorNoDevice ? DFX_ERR_NO_DEVICE : DFX_ERR_HIP, "%s: %s",   \
                  #expr, hipGetErrorString(e_));                                        \
  } while (0)

static size_t dt_size(int dt) { return (dt == DFX_F32 || dt == DFX_S32) ? 4 : 1; }

// ---- testing / tuning switches.  The environment is read ONCE, when the
//      library is first used; tests flip a switch afterwards with dfx_debug_set_tuning(). ----
namespace {
const char *const kTuningKeys[] = {"DFX_MAX_TH", "DFX_FORCE_GEOM", "DFX_STATIC_ROUNDS", "DFX_NO_FAST", "DFX_NO_MAGIC", "DFX_NO_LAZY", "DFX_HALF_UNITS", "DFX_NO_ROLES", "DFX_STORE_BOUND_BYTES",
                                   "DFX_STREAM_PXB", "DFX_STREAM_BLOCKING", "DFX_STREAM_PLANES", "DFX_STREAM_OCC_PAR",
                                   "DFX_STREAM_DIRECT", "DFX_STREAM_PW", "DFX_DIRECT_NPB", "DFX_DIRECT_NW", "DFX_DIRECT_WO1", "DFX_STREAM_GRID", "DFX_DEBUG_PTRS",
                                   "DEEPFUSION_PROFILE"};
struct Tuning {
  std::mutex mu;
  std::map<std::string, std::string> kv;
  Tuning() {
    for (const char *k : kTuningKeys)
      if (const char *v = getenv(k)) kv[k] = v;
  }
};
Tuning &tuning() {
  static Tuning t;
  return t;
}
// value of a switch or nullptr; the returned pointer stays valid until the switch is set again
// (thread_local copy: callers use it immediately)
const char *tune(const char *key) {
  static thread_local std::string held;
  Tuning &t = tuning();
  std::lock_guard<std::mutex> lk(t.mu);
  auto it = t.kv.find(key);
  if (it == t.kv.end()) return nullptr;
  held = it->second;
  return held.c_str();
}
}  // namespace

// launches of one MFMA-variant handle that may be in flight at the same time (any streams)
constexpr unsigned DFX_QUEUE_RING = 16;

// every entry point that takes a handle runs on the device the handle was created on
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (dev >= 0 && dev != prev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct dfx_conv {
  int device;  // ordinal the handle lives on (current device at dfx_conv_create)
  dfx_conv_desc d;
  int variant;
  ConvArgs args;
  MfmaGeom geom;
  StreamGeom sgeom;  // DFX_VARIANT_MFMA_STREAM
  DirectGeom dgeom;  // DFX_VARIANT_MFMA_STREAM served by conv_direct.cuh (fused ops): direct != 0
  int direct, nw, wo, wo1;
  int pw;            // DFX_VARIANT_MFMA_STREAM served by conv_pw.cuh (pointwise unfused): direct != 0 too (weights packed as for conv_direct.cuh)
  PwGeom pwgeom;
  int occ, pxb;      // stream variant: conv0 output blocks per chunk, pixel blocks per wave
  int icb, ocb, G, grid, block, lds;
  // role-specialised fused kernel (conv_mfma_roles.cuh): roles_ok = the SHAPE fits it (decided at create; LDS is
  // then sized for its larger control block), roles = the WEIGHTS allow its requant modes (decided by
  // dfx_conv_set_weights); otherwise the op runs on conv_mfma.cuh's kernel
  bool roles_ok, roles;
  void *d_wei, *d_wei1, *d_consts;
  int *d_queue;  // MFMA variant: ring of DFX_QUEUE_RING x {next unit, finished loaders}, one slot per launch in flight
  unsigned launch_seq;
  // In-flight guard of the queue ring.  Launches on ONE stream are ordered by the stream; as soon as a handle
  // has been submitted on a second stream every launch records its slot's event, and a launch that finds its
  // slot last used on another stream first waits (on the device) for that launch: a 17th concurrent launch
  // queues up behind the 1st instead of sharing its queue words.
  std::mutex *ring_mu;
  hipEvent_t slot_ev[DFX_QUEUE_RING];
  hipStream_t slot_stream[DFX_QUEUE_RING];
  unsigned char slot_state[DFX_QUEUE_RING];  // 0 never used, 1 used (no event recorded), 2 used + event recorded
  bool multi_stream;
  hipStream_t first_stream;
  int *trace_host;  // DFX_TRACE builds only
  unsigned long long *d_prof;  // DFX_STAMPS builds only
  bool weights_set;
  void *d_src, *d_dst;  // lazily allocated for dfx_conv_submit_host
  hipStream_t host_stream;
  char kernel_name[96];
};

struct dfx_concat {
  int device;
  dfx_concat_desc d;
  std::vector<int> channels;
  ConcatArgs args;
  std::vector<void *> d_srcs;  // lazily allocated for submit_host
  void *d_dst;
  hipStream_t host_stream;
};

extern "C" {

int dfx_version(void) { return DFX_VERSION; }
const char *dfx_last_error(void) { return g_err; }

int dfx_device_count(int *count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  if (count) *count = n;
  return DFX_OK;
}

int dfx_set_device(int ordinal) {
  HIP_TRY(hipSetDevice(ordinal));
  return DFX_OK;
}

int dfx_device_name(char *buf, size_t len) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, dev));
  snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return DFX_OK;
}

int dfx_mem_alloc_host(void **p, size_t bytes) {
  HIP_TRY(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
  return DFX_OK;
}
int dfx_mem_free_host(void *p) {
  if (p) HIP_TRY(hipHostFree(p));
  return DFX_OK;
}
int dfx_mem_alloc_device(void **p, size_t bytes) {
  HIP_TRY(hipMalloc(p, bytes ? bytes : 1));
  return DFX_OK;
}
int dfx_mem_free_device(void *p) {
  if (p) HIP_TRY(hipFree(p));
  return DFX_OK;
}
int dfx_memcpy_h2d(void *dst, const void *src, size_t bytes, dfx_stream_t s) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
  return DFX_OK;
}
int dfx_memcpy_d2h(void *dst, const void *src, size_t bytes, dfx_stream_t s) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
  return DFX_OK;
}
int dfx_memset_device(void *dst, int value, size_t bytes, dfx_stream_t s) {
  HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)s));
  return DFX_OK;
}
int dfx_stream_create(dfx_stream_t *s) {
  hipStream_t st;
  HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *s = st;
  return DFX_OK;
}
int dfx_stream_destroy(dfx_stream_t s) {
  if (s) HIP_TRY(hipStreamDestroy((hipStream_t)s));
  return DFX_OK;
}
int dfx_stream_sync(dfx_stream_t s) {
  HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return DFX_OK;
}
int dfx_stream_wait_stream(dfx_stream_t waiter, dfx_stream_t producer) {
  if (waiter == producer) return DFX_OK;
  // the event must live on the PRODUCER stream's device (the streams of a multi-device pipeline differ)
  int cur = -1, pdev = -1;
  (void)hipGetDevice(&cur);
  if (producer && hipStreamGetDevice((hipStream_t)producer, &pdev) != hipSuccess) pdev = -1;
  if (pdev >= 0 && pdev != cur) (void)hipSetDevice(pdev);
  hipEvent_t e = nullptr;
  hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (r == hipSuccess) r = hipEventRecord(e, (hipStream_t)producer);
  if (pdev >= 0 && pdev != cur) (void)hipSetDevice(cur);
  if (r == hipSuccess) r = hipStreamWaitEvent((hipStream_t)waiter, e, 0);
  if (e) (void)hipEventDestroy(e);  // (released once the recorded work has completed)
  HIP_TRY(r);
  return DFX_OK;
}
int dfx_event_create(dfx_event_t *e) {
  hipEvent_t ev;
  HIP_TRY(hipEventCreate(&ev));
  *e = ev;
  return DFX_OK;
}
int dfx_event_record(dfx_event_t e, dfx_stream_t s) {
  HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
  return DFX_OK;
}
int dfx_event_elapsed_ms(dfx_event_t start, dfx_event_t stop, float *ms) {
  HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DFX_OK;
}
int dfx_event_destroy(dfx_event_t e) {
  if (e) HIP_TRY(hipEventDestroy((hipEvent_t)e));
  return DFX_OK;
}

size_t dfx_blocked_offset(int o, int i, int kh, int kw, int I, int KH, int KW) {
  // [o/16][i/16][kh][kw][(i%16)/4][o%16][i%4], jit_conv_kernel.cc:333-338
  const size_t nb_ic = (size_t)I / 16;
  return ((((size_t)(o / 16) * nb_ic + (size_t)(i / 16)) * KH + kh) * KW + kw) * 256 +
         (size_t)((i % 16) / 4) * 64 + (size_t)(o % 16) * 4 + (size_t)(i % 4);
}

int dfx_reorder_oihw_to_blocked(const int8_t *oihw, int8_t *blocked, int O, int I, int KH,
                                int KW) {
  if (!oihw || !blocked || O <= 0 || I <= 0 || O % 16 || I % 16 || KH <= 0 || KW <= 0)
    return fail(DFX_ERR_INVALID, "reorder: O and I must be positive multiples of 16");
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int h = 0; h < KH; ++h)
        for (int w = 0; w < KW; ++w)
          blocked[dfx_blocked_offset(o, i, h, w, I, KH, KW)] =
              oihw[(((size_t)o * I + i) * KH + h) * KW + w];
  return DFX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// conv
// ---------------------------------------------------------------------------

static int validate_conv(const dfx_conv_desc &d) {
  auto is_dt = [](int v) { return v >= DFX_F32 && v <= DFX_U8; };
  if (d.bs <= 0 || d.ic <= 0 || d.oc <= 0 || d.ih <= 0 || d.iw <= 0 || d.kh <= 0 || d.kw <= 0 ||
      d.sh <= 0 || d.sw <= 0 || d.pad_t < 0 || d.pad_l < 0 || d.oc1x1 < 0)
    return fail(DFX_ERR_INVALID, "conv: non-positive dimension");
  if (!is_dt(d.dst_dt)) return fail(DFX_ERR_INVALID, "conv: bad dst dtype");  // jit_conv_kernel.cc:536-540
  if (d.bia0_dt != DFX_UNDEF && !is_dt(d.bia0_dt)) return fail(DFX_ERR_INVALID, "conv: bad bias dtype");
  if (d.bia1_dt != DFX_UNDEF && !is_dt(d.bia1_dt)) return fail(DFX_ERR_INVALID, "conv: bad bias1x1 dtype");
  if (d.ic % 16 || d.oc % 16)  // jit_conv_kernel.cc:590-592
    return fail(DFX_ERR_INVALID, "conv: ic and oc must be multiples of 16");
  if (d.oh != (d.ih + 2 * d.pad_t - d.kh) / d.sh + 1 || d.ow != (d.iw + 2 * d.pad_l - d.kw) / d.sw + 1 ||
      d.oh <= 0 || d.ow <= 0)  // op_conv.cc:290-297
    return fail(DFX_ERR_INVALID, "conv: output image size do not match");
  if (d.conv0_nscales != 1 && d.conv0_nscales != d.oc)  // op_conv.cc:311-313, :342-345
    return fail(DFX_ERR_INVALID, "conv: conv0 scales count must be 1 or oc");
  if (d.oc1x1) {
    if (d.oc1x1 % 16) return fail(DFX_ERR_INVALID, "conv: oc1x1 must be a multiple of 16");  // :619-621
    if (d.conv1_nscales != 1 && d.conv1_nscales != d.oc1x1)
      return fail(DFX_ERR_INVALID, "conv: conv1 scales count must be 1 or oc1x1");
  }
  if (d.conv0_round_mode < 0 || d.conv0_round_mode > 1 || d.conv1_round_mode < 0 || d.conv1_round_mode > 1)
    return fail(DFX_ERR_INVALID, "conv: bad round mode");
  return DFX_OK;
}

static size_t round16(size_t v) { return (v + 15) & ~(size_t)15; }

// Pick the unit decomposition of the MFMA variant: TH output rows x TW output
// columns per unit, either full-width rows (linear pixel numbering) or TW a
// multiple of 32.  Scored by how full a unit's 32-pixel tiles are (tiles rotate over
// the compute waves, so the tile count per unit need not match the wave count), how
// little halo it re-reads, whether the loader wave can hold the whole halo tile in
// registers, and -- for HBM-bound outputs -- how fine the dynamic hand-out is.
// false if nothing fits LDS.
// output bytes per pixel from which an op counts as bound by its output stream (finer units, lazy queue)
static size_t store_bound_bytes() {
  if (const char *e = tune("DFX_STORE_BOUND_BYTES")) return (size_t)std::max(1, atoi(e));  // tuning aid
  return 512;
}

static bool pick_geometry(const dfx_conv_desc &d, MfmaGeom &g, int &lds, int ctrl_bytes = MFMA_CTRL_BYTES) {
  const int ICB = d.ic / 32, OCB = d.oc / 32, NCB = d.oc1x1 / 32;
  const size_t fixed = (size_t)OCB * 9 * ICB * 1024 + (size_t)NCB * OCB * 1024 +
                       round16((size_t)mfma_cst_floats(d.oc, d.oc1x1) * 4) + (size_t)ctrl_bytes;
  const size_t lds_max = 163840;  // one workgroup per CU owns the whole LDS
  if (fixed + 4 * 1024 > lds_max) return false;
  const size_t tile_max = (lds_max - fixed) / MFMA_NB;  // ring of MFMA_NB tile slots
  // rows per unit wanted for parallelism: aim for >= ~3 units per team (512 teams)
  int th_par = (int)(((long long)d.bs * d.oh) / 1536);
  if (th_par < 1) th_par = 1;
  if (th_par < 2 && (long long)d.bs * d.oh / 2 >= 512) th_par = 2;
  if (th_par > 16) th_par = 16;
  if (const char *e = tune("DFX_MAX_TH")) th_par = std::max(1, std::min(th_par, atoi(e)));  // tuning aid
  const bool hbm_bound_dst = d.dst_dt == DFX_S32 || d.dst_dt == DFX_F32;
  double best = -1.0;
  for (int mode = 0; mode < 2; ++mode)
    for (int tw = (mode == 0 ? d.ow : 32); tw <= (mode == 0 ? d.ow : std::min(d.ow, 256)); tw += 32) {
      if (mode == 1 && tw >= d.ow) break;  // full width is mode 0
      for (int th = 1; th <= std::min(d.oh, th_par); ++th) {
        // a tile slot holds whole 1 KB pieces (64 granules of 16 bytes: one write per loader lane) + a dump piece
        const size_t tile = ((size_t)(th + 2) * (tw + 2) * (d.ic / 16) + 63) / 64 * 1024 + 1024;
        if (tile > tile_max) break;
        const int npx = th * tw;
        const int ntiles = mode == 0 ? (npx + 31) / 32 : th * (tw / 32);
        const double px_eff = (double)npx / (32.0 * ntiles);
        const double halo_eff = (double)npx / ((th + 2.0) * (tw + 2.0));
        // edge units are partial: fraction of the covered area that is real output
        const double cover = ((double)d.oh * d.ow) /
                             ((double)((d.oh + th - 1) / th * th) * ((d.ow + tw - 1) / tw * tw));
        const bool oversize = (tile - 1024) / 16 > (size_t)64 * MFMA_LC;
        double score = px_eff * cover * (0.75 + 0.25 * halo_eff) * (oversize ? 0.9 : 1.0);
        // The 14 compute waves of a CU claim tiles from the units in the 4-slot LDS ring: a unit should
        // bring >= 7 tiles so that two units in flight keep every wave busy while two more are staged
        // (measured at config 3, s32: 2-row units of 4 tiles 124 us, 4-row units of 7 tiles 97 us)
        const bool store_bound = hbm_bound_dst && (size_t)(d.oc1x1 > 0 ? d.oc1x1 : d.oc) * 4 >= store_bound_bytes();
        score *= 0.5 + 0.5 * std::min(1.0, ntiles / (store_bound ? 4.0 : 7.0));
        // Store-bound ops (>= 512 output bytes per pixel): workgroups drain at very different rates and
        // the lazy queue (conv_mfma.cuh) can only even that out with enough units per loader -- >= 6
        // (s32 headline, same box: 2-row units / 7 per loader 84.1 us, 4-row units / 3.5 per loader 86.5 us)
        if (store_bound) {
          if (mode == 1) score *= 0.9;  // column-split units end in partial tiles (the slower store form)
          const double units = (double)d.bs * ((d.oh + th - 1) / th) * ((d.ow + tw - 1) / tw);
          score *= 0.5 + 0.5 * std::min(1.0, units / (6.0 * 512.0));
        }
        if (score > best) {
          best = score;
          g.th = th; g.tw = tw; g.linear = mode == 0;
        }
      }
    }
  if (best < 0) return false;
  if (const char *e = tune("DFX_FORCE_GEOM")) {  // tuning aid: "th,tw" (must fit LDS)
    int fth = 0, ftw = 0;
    if (sscanf(e, "%d,%d", &fth, &ftw) == 2 && fth >= 1 && (ftw == d.ow || (ftw % 32 == 0 && ftw < d.ow)) &&
        ((size_t)(fth + 2) * (ftw + 2) * (d.ic / 16) + 63) / 64 * 1024 + 1024 <= tile_max) {
      g.th = fth; g.tw = ftw; g.linear = ftw == d.ow;
    }
  }
  g.pool = d.fuse_pool ? 1 : 0;
  if (g.pool) {
    // pooled ops: the scored unit (full-width rows, or a 32-multiple of columns) with an even number of rows
    // (a tile is 2 rows x 16 columns): the largest even th <= the scored one whose tile fits, at least 2
    auto tile_of = [&](int th, int tw) { return ((size_t)(th + 2) * (tw + 2) * (d.ic / 16) + 63) / 64 * 1024 + 1024; };
    int th = std::max(2, g.th & ~1);
    while (th > 2 && tile_of(th, g.tw) > tile_max) th -= 2;
    if (tile_of(th, g.tw) > tile_max) {  // rows of this width do not fit even in pairs: split the columns
      g.tw = 32 * std::max(1, std::min(d.ow / 32, 4));
      g.linear = g.tw == d.ow;
      while (g.tw > 32 && tile_of(th, g.tw) > tile_max) g.tw -= 32;
      if (tile_of(th, g.tw) > tile_max) return false;
    }
    g.th = th;
  }
  g.uy = (d.oh + g.th - 1) / g.th;
  g.ux = (d.ow + g.tw - 1) / g.tw;
  g.total_units = d.bs * g.uy * g.ux;
  g.row_chunks = (g.tw + 2) * (d.ic / 16);
  g.tile_chunks = (g.th + 2) * g.row_chunks;
  g.row_magic = (unsigned)(((1ull << 32) + g.row_chunks - 1) / g.row_chunks);
  g.tile_stride = (g.tile_chunks + 63) / 64 * 1024 + 1024;  // + the loader's dump piece
  g.tw_magic = (unsigned)(((1ull << 32) + g.tw - 1) / g.tw);
  g.upi_magic = g.uy * g.ux > 1 ? (unsigned)(((1ull << 32) + g.uy * g.ux - 1) / (g.uy * g.ux)) : 0u;
  g.ux_magic = g.ux > 1 ? (unsigned)(((1ull << 32) + g.ux - 1) / g.ux) : 0u;
  g.ntu = g.pool ? (g.th / 2) * ((g.tw + 15) / 16)
                 : g.linear ? (g.th * g.tw + 31) / 32 : g.th * (g.tw / 32);  // tile claims per unit
  g.ntu_magic = (unsigned)(((1ull << 32) + g.ntu - 1) / g.ntu);
  g.claim_limit = (int)std::min<long long>(0x7ffffff0LL, (long long)g.ntu * ((long long)g.total_units + 4));
  lds = (int)(fixed + (size_t)MFMA_NB * g.tile_stride);
  return true;
}

static bool mfma_eligible(const dfx_conv_desc &d) {  // fused or unfused (oc1x1 == 0)
  if ((long long)d.bs * d.oh * d.ow >= (1LL << 31) - 64) return false;  // (pixel indices are 32-bit in the kernel)
  return d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.pad_t <= 1 && d.pad_l <= 1 &&
         (d.ic == 32 || d.ic == 64) && (d.oc == 32 || d.oc == 64) && d.oc1x1 % 32 == 0;
}

// the role-specialised kernel (conv_mfma_roles.cuh): fused, 1-byte output, output channels in groups of 128 (<= 4)
static bool roles_eligible(const dfx_conv_desc &d) {
  if (tune("DFX_NO_ROLES") && atoi(tune("DFX_NO_ROLES")) != 0) return false;  // testing aid: conv_mfma.cuh's kernel
  // (must mirror DFX_FOR_EACH_ROLES_SHAPE in conv_mfma_roles_inst.inc: with oc = 64 only oc1x1 <= 256 is built --
  //  admitting 384 / 512 here made dfx_conv_create fail for those ops, found by profiles/debug/soak_resident.py)
  return mfma_eligible(d) && d.oc1x1 > 0 && !d.fuse_pool && (d.dst_dt == DFX_U8 || d.dst_dt == DFX_S8) &&
         d.oc1x1 % 128 == 0 && d.oc1x1 / 128 <= (d.oc == 64 ? 2 : 4);
}

// ---- direct-weight fused kernel (conv_direct.cuh) ----
static bool pick_direct_geometry(const dfx_conv_desc &d, int NW, int WO, int G, int npb, DirectGeom &g, int &lds) {
  const int M = 32 * npb;  // pixel slots per unit
  g.npb = npb;
  g.m0 = g.m1 = 0;
  const int ocb_real = (d.oc + 31) / 32;
  g.icb = (d.ic + 31) / 32;
  g.n_planes = (g.icb + 1) / 2;
  if (g.n_planes > 16) return false;  // (plane index 0..15: 4 bits in the staging table)
  g.ocb = (ocb_real + WO - 1) / WO * WO;
  g.n_g1 = ((d.oc1x1 + 31) / 32 + G - 1) / G;
  g.mid_stride = 32 * g.ocb + 16;
  const size_t cst_bytes = round16((size_t)3 * 32 * (g.ocb + G * g.n_g1) * 4);  // both stages' constants
  const size_t stage_bytes = (G == 4 && dt_size(d.dst_dt) == 1) ? (size_t)NW * DK_STAGE : 0;
  // (`mid`: the u8 intermediate / the 1-byte output rows; an unfused op with 4-byte output stages 32 px x 128 B per wave there)
  const size_t mid_bytes = std::max((size_t)M * g.mid_stride, (d.oc1x1 == 0 && dt_size(d.dst_dt) == 4) ? (size_t)NW * 32 * 144 : (size_t)0);
  const size_t fixed = 8 * M + mid_bytes + cst_bytes;  // (8 M: the slot tables pxoff and fboff)
  const size_t lds_max = 163840;
  g.unfused = d.oc1x1 == 0 ? 1 : 0;
  if ((long long)d.bs * d.oh * d.ow * (d.oc1x1 ? d.oc1x1 : d.oc) * (long long)dt_size(d.dst_dt) >= (1LL << 32) - 16 ||
      (long long)d.bs * d.oh * d.ow >= (1LL << 31) || (long long)d.ih * d.iw * d.ic * M >= (1LL << 31))
    return false;
  double best = -1.0;
  auto consider = [&](int ni, int thv, int twv) {
    const int lh = (thv - 1) * d.sh + d.kh, lw = (twv - 1) * d.sw + d.kw;
    if (lh >= 1024 || lw >= 1024 || ni >= 256) return;
    const long long npos = (long long)ni * lh * lw;
    // (+ 240 bytes per row and per image: the bank-spreading pads of the pitches, chosen below)
    const size_t tile = std::max((size_t)((long long)g.n_planes * ni * ((long long)lh * (lw * DK_POS + 240) + 240) + 16), stage_bytes);
    if (fixed + tile > lds_max) return;
    const double groups = (double)((d.bs + ni - 1) / ni);
    const double units = groups * ((d.oh + thv - 1) / thv) * ((d.ow + twv - 1) / twv);
    const double util = (double)d.bs * d.oh * d.ow / (units * M);
    const double halo = (double)thv * d.sh * twv * d.sw / ((double)lh * lw);
    const double two = fixed + tile <= 81920 ? 1.0 : 0.8;  // two workgroups per CU
    const double score = util * (0.8 + 0.2 * std::min(1.0, halo)) * two;
    if (score > best) {
      best = score;
      g.ni = ni; g.thv = thv; g.twv = twv; g.lh = lh; g.lw = lw; g.npos = (int)npos;
    }
  };
  if (d.oh * d.ow <= M) {
    for (int ni = std::min(d.bs, M / (d.oh * d.ow)); ni >= 1; --ni) consider(ni, d.oh, d.ow);
  } else {
    for (int twv = 1; twv <= std::min(d.ow, M); ++twv) consider(1, std::min(d.oh, M / twv), twv);
  }
  if (best < 0) return false;
  g.uy = (d.oh + g.thv - 1) / g.thv;
  g.ux = (d.ow + g.twv - 1) / g.twv;
  g.total_units = (d.bs + g.ni - 1) / g.ni * g.uy * g.ux;
  // Pitches: 16 c / 16 c' bytes of padding per tile row / image so that the 16 lanes of each ds_read_b128 lane
  // group ({0-3,12-15,20-27}, {4-11,16-19,28-31} of either half wave) read 16 different 16-byte bank columns --
  // brute force over c, c', cost = extra LDS cycles summed over the unit's pixel blocks (0 = conflict-free).
  {
    const int px_img = g.thv * g.twv, npx = g.ni * px_img;
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    long long best_cost = -1;
    int best_c = 0, best_ci = 0;
    for (int ci = 0; ci < (g.ni > 1 ? 16 : 1); ++ci)
      for (int c = 0; c < 16; ++c) {
        const long long rp = (long long)g.lw * DK_POS + 16 * c, ip = (long long)g.lh * rp + 16 * ci;
        long long cost = 0;
        for (int b = 0; b < npb; ++b)
          for (int gi = 0; gi < 2; ++gi) {
            int cnt[16] = {0};
            long long seen[16][16];
            for (int k = 0; k < 16; ++k) {
              const int pc = std::min(32 * b + grp[gi][k], npx - 1);
              const int img = pc / px_img, r = pc % px_img, ty = r / g.twv, tx = r % g.twv;
              const long long addr = img * ip + (long long)ty * d.sh * rp + (long long)tx * d.sw * DK_POS;
              const int col = (int)((addr / 16) % 16);
              bool dup = false;  // identical addresses broadcast
              for (int j = 0; j < cnt[col]; ++j) dup = dup || seen[col][j] == addr;
              if (!dup) seen[col][cnt[col]++] = addr;
            }
            int worst = 1;
            for (int k = 0; k < 16; ++k) worst = std::max(worst, cnt[k]);
            cost += worst - 1;
          }
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_c = c; best_ci = ci; }
      }
    g.row_pitch = g.lw * DK_POS + 16 * best_c;
    g.img_pitch = g.lh * g.row_pitch + 16 * best_ci;
  }
  g.plane_bytes = g.ni * g.img_pitch;
  g.off_pxoff = (int)round16(std::max((size_t)g.n_planes * (size_t)g.plane_bytes + 16, stage_bytes));
  g.off_mid = g.off_pxoff + 8 * M;
  g.off_cst = g.off_mid + (int)round16(mid_bytes);
  lds = g.off_cst + (int)cst_bytes;
  g.fast = 0;
  return true;
}

namespace dfx { int launch_conv_pw(const ConvArgs &, const PwGeom &, int, int, int, hipStream_t, int); }

static int direct_dispatch(dfx_conv *h, const ConvArgs &a, hipStream_t s, int mode) {
  if (h->pw) {  // pointwise unfused conv (conv_pw.cuh); the requant proofs live in dgeom (set_weights_direct)
    PwGeom pg = h->pwgeom;
    pg.fast = h->dgeom.fast;
    pg.m0 = h->dgeom.m0;
    return dfx::launch_conv_pw(a, pg, h->d.dst_dt, h->grid, h->lds, s, mode);
  }
  switch (h->d.dst_dt) {
    case DFX_F32: return launch_conv_direct_f32(a, h->dgeom, h->nw, h->wo, h->G, h->wo1, h->grid, h->lds, s, mode);
    case DFX_S32: return launch_conv_direct_s32(a, h->dgeom, h->nw, h->wo, h->G, h->wo1, h->grid, h->lds, s, mode);
    case DFX_S8: return launch_conv_direct_s8(a, h->dgeom, h->nw, h->wo, h->G, h->wo1, h->grid, h->lds, s, mode);
    case DFX_U8: return launch_conv_direct_u8(a, h->dgeom, h->nw, h->wo, h->G, h->wo1, h->grid, h->lds, s, mode);
  }
  return -1;
}

// ---- streamed-weight variant (conv_stream.cuh) ----
// blocks per chunk / group: the largest of {4,2,1} that pads the block count by <= 1/3
static int pick_blocking(int nblocks) {
  for (int x : {4, 2, 1}) {
    const int padded = (nblocks + x - 1) / x * x;
    if (3 * padded <= 4 * nblocks) return x;
  }
  return 1;
}

// chunk_par: the op will hand out (unit, output chunk) items (one output chunk per item)
static bool pick_stream_geometry(const dfx_conv_desc &d, int OCC, int G, int PXB, bool chunk_par, StreamGeom &g,
                                 int &lds) {
  const int M = ST_M * PXB;  // pixel slots per unit
  const bool fused = d.oc1x1 > 0;
  const int icb = (d.ic + 31) / 32, ocb_real = (d.oc + 31) / 32;
  g.icb = icb;
  g.n_icc = (icb + 1) / 2;
  g.n_occ = (ocb_real + OCC - 1) / OCC;
  g.ocb = g.n_occ * OCC;
  g.n_g1 = fused ? ((d.oc1x1 + 31) / 32 + G - 1) / G : 0;
  g.ks2 = (g.ocb + 1) / 2;
  g.mid_stride = 32 * g.ocb + 16;
  int s0 = 0;
  for (int c = 0; c < g.n_icc; ++c) s0 += (d.kh * d.kw * std::min(2, icb - 2 * c) + 1) / 2;
  g.s0_steps = s0 * g.n_occ;
  const int WB = fused ? std::max(OCC, G) : OCC;
  const size_t cst_bytes = fused ? round16((size_t)3 * 32 * g.ocb * 4) : 0;
  // 1-byte store staging, 4 waves (fused: 4-block groups; unfused: chunks of >= 2 blocks)
  const size_t stage_bytes = (dt_size(d.dst_dt) == 1 && (fused ? G == 4 : OCC >= 2)) ? (size_t)4 * ST_STAGE : 0;
  const size_t fixed = (size_t)3 * 2 * WB * 1024 + 4 * M + (fused ? (size_t)M * g.mid_stride : 0) + cst_bytes;
  const size_t lds_max = 163840;
  // index ranges the kernel keeps in 32 bits / packed fields
  // (dst byte offsets are 32-bit in the kernel's pixel table)
  if ((long long)d.bs * d.oh * d.ow * (d.oc1x1 ? d.oc1x1 : d.oc) * (long long)dt_size(d.dst_dt) >= (1LL << 32) - 16 ||
      (long long)d.bs * d.oh * d.ow >= (1LL << 31) || (long long)d.ih * d.iw * d.ic * M >= (1LL << 31)) return false;
  double best = -1.0;
  auto consider = [&](int ni, int thv, int twv) {
    const int lh = (thv - 1) * d.sh + d.kh, lw = (twv - 1) * d.sw + d.kw;
    if (lh >= 1024 || lw >= 1024) return;
    const long long npos = (long long)ni * lh * lw;
    if (fixed + std::max((size_t)npos * ST_POS + 16, stage_bytes) > lds_max) return;
    const double groups = (double)((d.bs + ni - 1) / ni);
    const double units = groups * ((d.oh + thv - 1) / thv) * ((d.ow + twv - 1) / twv);
    const double util = (double)d.bs * d.oh * d.ow / (units * M);           // filled pixel slots
    const double halo = (double)thv * d.sh * twv * d.sw / ((double)lh * lw);  // input re-read
    const double score = util * (0.8 + 0.2 * std::min(1.0, halo));
    if (score > best) {
      best = score;
      g.ni = ni; g.thv = thv; g.twv = twv; g.lh = lh; g.lw = lw; g.npos = (int)npos;
    }
  };
  if (d.oh * d.ow <= M) {
    for (int ni = std::min(d.bs, M / (d.oh * d.ow)); ni >= 1; --ni) consider(ni, d.oh, d.ow);
  } else {
    for (int twv = 1; twv <= std::min(d.ow, M); ++twv) consider(1, std::min(d.oh, M / twv), twv);
  }
  if (best < 0) return false;
  g.uy = (d.oh + g.thv - 1) / g.thv;
  g.ux = (d.ow + g.twv - 1) / g.twv;
  g.total_units = (d.bs + g.ni - 1) / g.ni * g.uy * g.ux;
  g.off_tile = 3 * 2 * WB * 1024;  // three weight buffers
  // All input chunks resident in LDS (staged once per work item, without register prefetch)
  // instead of one chunk at a time (prefetched): pays when a work item would otherwise stage
  // every chunk once per output chunk, or when a chunk has too few steps to hide the next
  // fetch (1x1 kernels) -- if that still leaves room for two workgroups per CU.
  g.planes = 1;
  const bool restaged = !chunk_par && g.n_occ > 1;
  const bool short_chunks = d.kh * d.kw <= 2;
  if (g.n_icc > 1 && (restaged || short_chunks) &&
      fixed + std::max((size_t)g.n_icc * g.npos * ST_POS + 16, stage_bytes) <= 81920)
    g.planes = g.n_icc;
  if (const char *e = tune("DFX_STREAM_PLANES"))  // testing aid: 0 = never, 1 = whenever it fits LDS at all
    g.planes = (atoi(e) && g.n_icc > 1 && fixed + std::max((size_t)g.n_icc * g.npos * ST_POS + 16, stage_bytes) <= lds_max) ? g.n_icc : 1;
  {
    auto magic = [](long long x) { return (unsigned)(((1ull << 32) + (unsigned long long)x - 1) / (unsigned long long)x); };
    g.mg_g4 = magic(4 * g.planes); g.mg_lhw = magic((long long)g.lh * g.lw); g.mg_lw = magic(g.lw);
  }
  // +16: the staging dump slot; the 1-byte store staging areas alias the tile
  g.off_pxoff = (int)round16((size_t)g.off_tile + std::max((size_t)g.planes * g.npos * ST_POS + 16, stage_bytes));
  g.off_mid = g.off_pxoff + 4 * M;
  g.off_cst = g.off_mid + (fused ? M * g.mid_stride : 0);
  g.off_stage = g.off_tile;
  lds = g.off_cst + (int)cst_bytes;
  return true;
}

static int stream_dispatch(dfx_conv *h, const ConvArgs &a, hipStream_t s, int mode) {
  const int fused = h->d.oc1x1 > 0;
  switch (h->d.dst_dt) {
    case DFX_F32: return launch_conv_stream_f32(a, h->sgeom, h->occ, h->G, h->pxb, fused, h->grid, h->lds, s, mode);
    case DFX_S32: return launch_conv_stream_s32(a, h->sgeom, h->occ, h->G, h->pxb, fused, h->grid, h->lds, s, mode);
    case DFX_S8: return launch_conv_stream_s8(a, h->sgeom, h->occ, h->G, h->pxb, fused, h->grid, h->lds, s, mode);
    case DFX_U8: return launch_conv_stream_u8(a, h->sgeom, h->occ, h->G, h->pxb, fused, h->grid, h->lds, s, mode);
  }
  return -1;
}

// `a` and `g` are per-launch copies (src/dst pointers, queue slot): submits of one handle do not
// share mutable host state
static int mfma_dispatch(dfx_conv *h, const ConvArgs &a, const MfmaGeom &g, hipStream_t s, int mode) {
  if (h->variant == DFX_VARIANT_MFMA_STREAM) return h->direct ? direct_dispatch(h, a, s, mode) : stream_dispatch(h, a, s, mode);
  if (h->variant == DFX_VARIANT_MFMA_CONV) {
    switch (h->d.dst_dt) {
      case DFX_F32: return launch_conv_mfma_f32_unfused(a, g, h->icb, h->ocb, h->grid, h->lds, s, mode);
      case DFX_S32: return launch_conv_mfma_s32_unfused(a, g, h->icb, h->ocb, h->grid, h->lds, s, mode);
      case DFX_S8: return launch_conv_mfma_s8_unfused(a, g, h->icb, h->ocb, h->grid, h->lds, s, mode);
      case DFX_U8: return launch_conv_mfma_u8_unfused(a, g, h->icb, h->ocb, h->grid, h->lds, s, mode);
    }
    return -1;
  }
  if (h->roles_ok && (mode == 1 || h->roles)) {
    const int ncb = h->d.oc1x1 / 32;
    const int rc = h->d.dst_dt == DFX_U8 ? launch_conv_mfma_roles_u8(a, g, h->icb, h->ocb, ncb, h->grid, h->lds, s, mode)
                                         : launch_conv_mfma_roles_s8(a, g, h->icb, h->ocb, ncb, h->grid, h->lds, s, mode);
    if (mode != 1 || rc != 0) return rc;  // (mode 1 raises the LDS limit of BOTH kernels)
  }
  switch (h->d.dst_dt) {
    case DFX_F32: return launch_conv_mfma_f32(a, g, h->icb, h->ocb, h->G, h->grid, h->lds, s, mode);
    case DFX_S32: return launch_conv_mfma_s32(a, g, h->icb, h->ocb, h->G, h->grid, h->lds, s, mode);
    case DFX_S8: return launch_conv_mfma_s8(a, g, h->icb, h->ocb, h->G, h->grid, h->lds, s, mode);
    case DFX_U8: return launch_conv_mfma_u8(a, g, h->icb, h->ocb, h->G, h->grid, h->lds, s, mode);
  }
  return -1;
}

// releases everything a conv handle owns (also used on dfx_conv_create's failure paths)
static void conv_release(dfx_conv *h) {
  if (!h) return;
  DeviceGuard dg(h->device);
  if (h->host_stream) (void)hipStreamDestroy(h->host_stream);
  (void)hipFree(h->d_wei); (void)hipFree(h->d_wei1); (void)hipFree(h->d_consts);
  (void)hipFree(h->d_src); (void)hipFree(h->d_dst); (void)hipFree(h->d_queue);
  (void)hipFree(h->d_prof);
  for (unsigned i = 0; i < DFX_QUEUE_RING; ++i)
    if (h->slot_ev[i]) (void)hipEventDestroy(h->slot_ev[i]);
  delete h->ring_mu;
  delete h;
}

extern "C" {

int dfx_conv_create(const dfx_conv_desc *desc, dfx_conv_t **out) {
  if (!desc || !out) return fail(DFX_ERR_INVALID, "conv_create: null argument");
  *out = nullptr;
  int rc = validate_conv(*desc);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(DFX_ERR_NO_DEVICE, "conv_create: no HIP device (this library has no CPU path)");

  dfx_conv *h = new (std::nothrow) dfx_conv();
  if (!h) return fail(DFX_ERR_HIP, "out of host memory");
  memset(static_cast<void *>(h), 0, sizeof(*h));
  h->d = *desc;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  const dfx_conv_desc &d = h->d;

  ConvArgs &a = h->args;
  a.bs = d.bs; a.ic = d.ic; a.ih = d.ih; a.iw = d.iw; a.oc = d.oc; a.oh = d.oh; a.ow = d.ow;
  a.kh = d.kh; a.kw = d.kw; a.sh = d.sh; a.sw = d.sw; a.pt = d.pad_t; a.pl = d.pad_l;
  a.oc1 = d.oc1x1; a.dst_dt = d.dst_dt; a.relu0 = d.conv0_relu; a.relu1 = d.conv1_relu;
  a.rm0 = d.conv0_round_mode; a.rm1 = d.conv1_round_mode;

  bool want_mfma = mfma_eligible(d) && d.force_variant != DFX_VARIANT_GENERIC &&
                   d.force_variant != DFX_VARIANT_MFMA_STREAM;
  if (d.fuse_pool) {  // fused 2x2/2 max pooling: the resident-weight kernel's unfused form only
    if (d.fuse_pool != 2 || d.oc1x1 != 0 || !want_mfma || (d.oh & 1) || (d.ow & 1)) {
      conv_release(h);
      return fail(DFX_ERR_UNSUPPORTED, "conv_create: fused pooling needs an unfused 3x3 stride-1 conv with 32/64 channels and even output size");
    }
  }
  if ((d.force_variant == DFX_VARIANT_MFMA_FUSED || d.force_variant == DFX_VARIANT_MFMA_CONV) && !mfma_eligible(d)) {
    conv_release(h);
    return fail(DFX_ERR_UNSUPPORTED, "conv_create: shape not covered by the MFMA variant");
  }
  const bool want_stream = !want_mfma && d.force_variant != DFX_VARIANT_GENERIC;
  bool stream_ok = false;
  if (want_stream) {
    h->occ = pick_blocking((d.oc + 31) / 32);
    h->G = d.oc1x1 ? pick_blocking((d.oc1x1 + 31) / 32) : h->occ;
    if (const char *e = tune("DFX_STREAM_BLOCKING")) {  // tuning aid: "occ,g" from {1,2,4}
      int o = 0, gg = 0;
      if (sscanf(e, "%d,%d", &o, &gg) == 2 && (o == 1 || o == 2 || o == 4) && (gg == 1 || gg == 2 || gg == 4)) {
        h->occ = o;
        h->G = d.oc1x1 ? gg : o;
      }
    }
    // Two pixel blocks per wave halve the LDS fragment traffic per MFMA but double the unit:
    // taken when the units still fill the machine twice over and two workgroups still share a CU.
    int ncu = 256;
    {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    }
    h->pxb = 2;
    if (const char *e = tune("DFX_STREAM_PXB")) h->pxb = atoi(e) == 1 ? 1 : 2;  // tuning aid
    stream_ok = h->pxb == 2 && pick_stream_geometry(d, h->occ, h->G, 2, false, h->sgeom, h->lds) &&
                ((h->sgeom.total_units >= 4 * ncu && h->lds <= 81920) || tune("DFX_STREAM_PXB"));
    if (!stream_ok) {
      h->pxb = 1;
      stream_ok = pick_stream_geometry(d, h->occ, h->G, 1, false, h->sgeom, h->lds);
    }
    if (!stream_ok && d.force_variant == DFX_VARIANT_MFMA_STREAM) {
      conv_release(h);
      return fail(DFX_ERR_UNSUPPORTED, "conv_create: shape does not fit the streamed MFMA variant");
    }
  }
  // The direct-weight kernel (conv_direct.cuh) serves fused ops with >= 64 channels on both sides wherever one of
  // its instances fits (N=128, u8 out, against conv_stream.cuh: res3 37 vs 50 us, res4 31 vs 53, res5 43 vs 79
  // in two launches, res3s2 45 vs 63: profiles/r03/direct_sweep_*.txt).  DFX_STREAM_DIRECT=0 turns it off.
  // Unfused ops (late round 3): the same kernel without its 1x1 stage (DirectGeom::unfused), for convs with a window
  // (kh * kw > 1) and >= 64 channels on both sides -- VGG conv3 / conv5-style layers.  Pointwise convs stay on
  // conv_stream.cuh (HBM-bound by their input; pw256 measured here 55 us against 42, profiles/r03/unfused_pointwise.txt).
  const bool direct_fused = d.oc1x1 > 0 && d.oc >= 64 && d.oc1x1 >= 64;
  // ... except deep ones whose weights do not fit conv_pw.cuh's LDS image (ic >= 512: res4 / res5 reduce convs, 1024 ->
  // 256 at 14 x 14: 26.8 us here against 31.3, profiles/r03/unfused_pointwise.txt).
  const bool direct_unfused = d.oc1x1 == 0 && d.oc >= 64 && d.ic >= 64 && (d.kh * d.kw > 1 || d.ic >= 512) && !d.fuse_pool;
  // Pointwise unfused convs whose weights fit LDS: conv_pw.cuh (pixel fragments straight from global memory into the
  // MFMA operands, no input tile).  DFX_STREAM_PW=0 turns it off (those shapes then run on conv_stream.cuh).
  bool want_pw = stream_ok && d.oc1x1 == 0 && d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.pad_t == 0 && d.pad_l == 0 &&
                 !d.fuse_pool && d.ic % 256 == 0 && (d.oc == 64 || d.oc == 128 || d.oc == 256) && (long long)d.oc * d.ic <= 98304 &&
                 (long long)d.bs * d.oh * d.ow < (1LL << 31) - 64;
  if (const char *e = tune("DFX_STREAM_PW")) want_pw = want_pw && atoi(e) != 0;
  if (want_pw) {
    memset(&h->dgeom, 0, sizeof(h->dgeom));
    h->dgeom.icb = d.ic / 32;
    h->dgeom.ocb = d.oc / 32;
    h->dgeom.n_g1 = 0;
    h->dgeom.unfused = 1;
    h->dgeom.npb = 1;
    h->G = 4;
    h->pw = 1;
    h->direct = 1;
    h->nw = PW_THREADS / 64;
    h->pwgeom.icb = h->dgeom.icb;
    h->pwgeom.ocb = h->dgeom.ocb;
    h->pwgeom.px_total = d.bs * d.oh * d.ow;
    h->pwgeom.n_blocks = (h->pwgeom.px_total + 31) / 32;
    h->pwgeom.off_cst = d.oc * d.ic;
    h->pwgeom.fast = h->pwgeom.m0 = 0;
    h->pwgeom.off_stage = (int)round16((size_t)d.oc * d.ic + (size_t)3 * d.oc * 4);
    h->pwgeom.stage_bytes = (int)round16(dt_size(d.dst_dt) == 1 ? (size_t)32 * (d.oc + 16) : (size_t)32 * 144);
    h->lds = h->pwgeom.off_stage + h->nw * h->pwgeom.stage_bytes;
    h->dgeom.total_units = (h->pwgeom.n_blocks + h->nw - 1) / h->nw;  // (workgroups' worth of blocks: the grid's upper bound)
    h->dgeom.thv = 0; h->dgeom.uy = h->dgeom.ux = 1;
  }
  bool want_direct = stream_ok && !h->pw && (direct_fused || direct_unfused);
  if (const char *e = tune("DFX_STREAM_DIRECT")) want_direct = want_direct && atoi(e) != 0;
  if (want_direct) {
    const int ocb2 = ((d.oc + 31) / 32 + 1) / 2 * 2;
    const int ncb1 = (d.oc1x1 + 31) / 32;
    const int G = direct_unfused ? 4 : (ncb1 >= 3 ? 4 : 2);  // (unfused: the instances with G = 4 and no 1x1 groups)
    // Pixel blocks per unit: the largest of 4, 2, 1 whose units still fill three quarters of the workgroup slots
    // (two per CU).  Smaller units re-stream the weights more often, larger ones leave CUs (res4: 196 units of
    // 128 pixels for 512 slots, res5: 64) or SIMDs empty.  With fewer than 4 blocks the four waves split the
    // output blocks of BOTH stages four ways (WO = WO1 = 4).  DFX_DIRECT_NPB forces one (testing aid).
    int ncu2 = 512;
    {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu2 = 2 * prop.multiProcessorCount;
    }
    int forced = 0, forced_nw = 0, forced_wo1 = 0;
    if (const char *e = tune("DFX_DIRECT_NPB")) forced = atoi(e);
    if (const char *e = tune("DFX_DIRECT_NW")) forced_nw = atoi(e);
    if (const char *e = tune("DFX_DIRECT_WO1")) forced_wo1 = atoi(e);
    const int ocb_real = (d.oc + 31) / 32, n_g1 = (ncb1 + G - 1) / G;
    // candidates in order of preference (instances: conv_direct_inst.inc):
    //  * four waves, 128-pixel units, where those fill three quarters of the workgroup slots (two per CU);
    //  * eight waves (one workgroup per CU, two waves per SIMD, each weight byte fetched once or twice per
    //    unit) with the largest unit that still fills three quarters of the CUs;
    //  * four waves with smaller units.
    struct Cand { int nw, wo, wo1, npb, min_units; };
    std::vector<Cand> cands;
    const int wo4 = ocb2 % 4 == 0 ? 4 : 2;
    const int wo1_4 = (G == 4 && wo4 == 4 && n_g1 % 4 == 0 && !direct_unfused) ? 4 : 1;
    cands.push_back({4, wo4, wo1_4, 4, 3 * ncu2 / 4});
    if (wo1_4 == 4) cands.push_back({4, wo4, 1, 4, 3 * ncu2 / 4});
    if (G == 4 && ocb_real % 8 == 0) {
      // (a pointwise conv's units are short -- one tap -- and 64-pixel units measured better than 128: tried first)
      if (direct_unfused && d.kh * d.kw == 1) cands.push_back({8, 8, 8, 2, 3 * ncu2 / 8});
      for (int npb : {4, 2, 1}) cands.push_back({8, 8, 8, npb, npb == 1 ? 0 : 3 * ncu2 / 8});
    }
    for (int npb : {2, 1}) cands.push_back({4, 4, 4, npb, npb == 1 ? 0 : 3 * ncu2 / 4});
    // first pass: four-wave candidates must also leave room for two workgroups per CU (80 KB of LDS each; a
    // stride-2 layer's 128-pixel halo tile does not: res3s2 72 us with 128-pixel units, 45 us with 64)
    // ... and every candidate must fill its pixel slots: where the LDS only admits a sliver of a patch (512 input
    // channels at 14 x 14: 14 x 3 pixels of a 128-slot unit) the next smaller unit is the better kernel (vgg5
    // unfused: 95 us with 128-slot units, 39 us with 64).  Passes: 0 = all rules, 1 = without the LDS rule,
    // 2 = anything that fits.
    for (int pass = 0; pass < 3 && !h->direct; ++pass)
      for (const Cand &c : cands) {
        if (forced && c.npb != forced) continue;
        if (forced_nw && c.nw != forced_nw) continue;
        if (forced_wo1 && c.wo1 != forced_wo1) continue;
        DirectGeom dg;
        memset(&dg, 0, sizeof(dg));
        int lds = 0;
        if (!pick_direct_geometry(d, c.nw, c.wo, G, c.npb, dg, lds)) continue;
        if (!forced && dg.total_units < c.min_units) continue;  // too few units: try the next candidate
        if (!forced && pass == 0 && c.nw == 4 && lds > 81920) continue;
        const double util = (double)d.bs * d.oh * d.ow / ((double)dg.total_units * 32.0 * c.npb);
        if (!forced && pass < 2 && util < 0.7) continue;
        if (tune("DFX_DEBUG_PTRS"))
          fprintf(stderr, "[dfx] direct candidate nw%d wo%d wo1 %d npb%d: ni %d th %d tw %d units %d (min %d) lds %d pass %d\n", c.nw, c.wo, c.wo1,
                  c.npb, dg.ni, dg.thv, dg.twv, dg.total_units, c.min_units, lds, pass);
        h->dgeom = dg;
        h->direct = 1;
        h->G = G;
        h->nw = c.nw;
        h->wo = c.wo;
        h->wo1 = c.wo1;
        h->lds = lds;
        break;
      }
  }
  if (stream_ok && h->direct) {
    h->variant = DFX_VARIANT_MFMA_STREAM;
    h->block = 64 * h->nw;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot query the device");
    }
    if (mfma_dispatch(h, h->args, h->geom, nullptr, 1) != 0) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot raise dynamic LDS limit to %d bytes", h->lds);
    }
    int per_cu = mfma_dispatch(h, h->args, h->geom, nullptr, 2);
    if (per_cu < 1) per_cu = 1;
    h->grid = std::min(h->dgeom.total_units, prop.multiProcessorCount * per_cu);
    if (const char *e = tune("DFX_STREAM_GRID")) h->grid = std::max(1, std::min(h->grid, atoi(e)));  // testing aid
    a.rows_per_unit = h->dgeom.thv;
    a.units_per_image = h->dgeom.uy * h->dgeom.ux;
    if (h->pw)
      snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_pw_kernel<%d,%d>", h->dgeom.ocb, d.dst_dt);
    else
      snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_direct_kernel<nw%d,%d,%d,%d,%d,%snpb%d>", h->nw, h->wo, h->G, h->wo1, d.dst_dt,
               h->dgeom.unfused ? "unfused," : "", h->dgeom.npb);
#ifdef DFX_STAMPS
    if (hipMalloc((void **)&h->d_prof, (size_t)h->grid * h->nw * 16 * 8) != hipSuccess || hipMemset(h->d_prof, 0, (size_t)h->grid * h->nw * 16 * 8) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot allocate the stamp buffer");
    }
    h->dgeom.prof = h->d_prof;
#endif
#ifdef DK_DEBUG
    if (hipMalloc((void **)&h->d_prof, 64 * 8) != hipSuccess || hipMemset(h->d_prof, 0xff, 64 * 8) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot allocate the debug buffer");
    }
    h->dgeom.dbg = (long long *)h->d_prof;
    h->dgeom.src_bytes = (long long)d.bs * d.ih * d.iw * d.ic;
    h->dgeom.dst_bytes = (long long)d.bs * d.oh * d.ow * (d.oc1x1 ? d.oc1x1 : d.oc) * (long long)dt_size(d.dst_dt);
#endif
  } else if (stream_ok) {
    h->variant = DFX_VARIANT_MFMA_STREAM;
    h->block = ST_THREADS;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot query the device");
    }
    if (mfma_dispatch(h, h->args, h->geom, nullptr, 1) != 0) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot raise dynamic LDS limit to %d bytes", h->lds);
    }
    int per_cu = mfma_dispatch(h, h->args, h->geom, nullptr, 2);
    if (per_cu < 1) per_cu = 1;
    const int capacity = prop.multiProcessorCount * per_cu;  // resident workgroups
    // unfused op whose units leave workgroup slots empty: hand out (unit, output chunk) items
    h->sgeom.occ_par = (d.oc1x1 == 0 && h->sgeom.n_occ > 1 && h->sgeom.total_units < 2 * capacity) ? 1 : 0;
    if (const char *e = tune("DFX_STREAM_OCC_PAR")) h->sgeom.occ_par = (d.oc1x1 == 0 && h->sgeom.n_occ > 1 && atoi(e)) ? 1 : 0;  // testing aid
    if (h->sgeom.occ_par) {  // lay LDS out again for one output chunk per item (same units)
      const int units = h->sgeom.total_units;
      if (!pick_stream_geometry(d, h->occ, h->G, h->pxb, true, h->sgeom, h->lds) || h->sgeom.total_units != units ||
          mfma_dispatch(h, h->args, h->geom, nullptr, 1) != 0) {
        conv_release(h);
        return fail(DFX_ERR_HIP, "conv_create: internal: chunk-parallel layout failed");
      }
      h->sgeom.occ_par = 1;
    }
    h->grid = std::min(h->sgeom.total_units * (h->sgeom.occ_par ? h->sgeom.n_occ : 1), capacity);
    // (Until round 3 a fused op with too few units to fill the machine ran as two launches through an
    // intermediate in global memory; conv_direct.cuh's one-launch kernel replaced that path: res5 79 -> 43 us.)
    if (const char *e = tune("DFX_STREAM_GRID")) h->grid = std::max(1, std::min(h->grid, atoi(e)));  // testing aid
#ifdef DFX_STAMPS
    if (hipMalloc((void **)&h->d_prof, (size_t)h->grid * 96 * 8) != hipSuccess ||
        hipMemset(h->d_prof, 0, (size_t)h->grid * 96 * 8) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot allocate the stamp buffer");
    }
    h->sgeom.prof = h->d_prof;
#endif
    a.rows_per_unit = h->sgeom.thv;
    a.units_per_image = h->sgeom.uy * h->sgeom.ux;
    snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_stream_kernel<%d,%d,%d,%d,%s%s>", h->occ, h->G, h->pxb,
             d.dst_dt, d.oc1x1 ? "fused" : "unfused", h->sgeom.occ_par ? ",occ_par" : "");
  } else if (d.fuse_pool && !(want_mfma && pick_geometry(d, h->geom, h->lds))) {
    conv_release(h);  // (no other kernel knows about the pooled destination)
    return fail(DFX_ERR_UNSUPPORTED, "conv_create: no unit geometry of the resident-weight kernel fits fused pooling here");
  } else if (want_mfma && pick_geometry(d, h->geom, h->lds, roles_eligible(d) ? RL_CTRL_BYTES : MFMA_CTRL_BYTES)) {
    const bool fused = d.oc1x1 > 0;
    h->roles_ok = roles_eligible(d);
    h->variant = fused ? DFX_VARIANT_MFMA_FUSED : DFX_VARIANT_MFMA_CONV;
    h->icb = d.ic / 32; h->ocb = d.oc / 32;
    const int ncb = d.oc1x1 / 32;
    h->G = fused ? ((ncb % 4 == 0) ? 4 : (ncb % 2 == 0 ? 2 : 1)) : h->ocb;
    {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        conv_release(h);
        return fail(DFX_ERR_HIP, "conv_create: cannot query the device");
      }
      h->grid = prop.multiProcessorCount;  // one persistent workgroup (two teams) per CU
      const int want = (h->geom.total_units + MFMA_TEAMS - 1) / MFMA_TEAMS;
      if (h->grid > want) h->grid = want;
    }
    h->block = MFMA_THREADS;
    if (hipMalloc((void **)&h->d_queue, 8 * DFX_QUEUE_RING) != hipSuccess ||
        hipMemset(h->d_queue, 0, 8 * DFX_QUEUE_RING) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot allocate the unit queue");
    }
    h->geom.queue = h->d_queue;
    h->ring_mu = new (std::nothrow) std::mutex();
    if (!h->ring_mu) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "out of host memory");
    }
    h->geom.mode0 = h->geom.mode1 = 0;
    {
      const int teams = h->grid * MFMA_TEAMS;
      // Few units per loader (the headline block: 3.5): the queue's granularity -- a whole unit, half a
      // round of the 14 compute waves -- makes workgroups end up with 6..8 units and the slowest sets the
      // kernel time (measured spread of workgroup lifetimes: 30 %).  A static split (stream-major, see
      // stream_id in conv_mfma.cuh) gives every workgroup the same count +- 1 unit.  Many units per
      // loader: three static rounds, then the queue evens out speed differences between CUs.
      // (Also for the store-bound s32 headline, whose XCDs finish 40 % apart: static 82.5 us median,
      // two static rounds + queue 88.5 us, profiles/r02/ab_s32.txt.)
      const int rounds = (h->geom.total_units + teams - 1) / teams;
      h->geom.static_rounds = rounds <= 8 ? rounds : 3;
      // store-bound op (>= 512 output bytes per pixel): two static rounds (staged before the barrier),
      // then the queue with lazy draws -- see the loader in conv_mfma.cuh
      const size_t out_px_bytes = (size_t)(d.oc1x1 > 0 ? d.oc1x1 : d.oc) * dt_size(d.dst_dt);
      h->geom.lazy_queue = 0;
      if (out_px_bytes >= store_bound_bytes() && rounds > 2 && !tune("DFX_NO_LAZY")) {
        h->geom.lazy_queue = 1;
        h->geom.static_rounds = 2;
      }
      if (const char *e = tune("DFX_STATIC_ROUNDS")) h->geom.static_rounds = std::min(rounds, std::max(0, atoi(e)));  // tuning aid
      // Store-bound ops, the tail: a loader stream works through a unit in ~1/7 of the launch (headline: 10.7 us)
      // and a workgroup still holds up to four when the queue runs dry, so workgroups finish up to ~20 us apart
      // (stamps: a CU is idle 8.4 us = 10.5 % of the span before the kernel ends).  Handing out the last units as
      // two half units each (ids >= half_from, MfmaGeom::half_from: the same tiles, the same bytes, half the
      // granule) pays only where a HALF unit still brings >= 7 tiles for the 14 compute waves: VGG conv1_2 f32
      // (224-pixel rows) 325.5-325.9 us with the last 2048 units split against 327.2-328.6; the headline's
      // half units have 2 tiles and cost it 0.3 % (256 units split) to 13 % (2048): profiles/r03/ab_half_units.txt.
      // DFX_HALF_UNITS = number of units to split (0: none) overrides the rule.
      h->geom.half_from = 0x7fffffff;
      if (h->geom.lazy_queue && !h->geom.pool && h->geom.th % 2 == 0) {
        const int half_tiles = h->geom.linear ? ((h->geom.th / 2) * h->geom.tw + 31) / 32 : (h->geom.th / 2) * (h->geom.tw / 32);
        long long nh = half_tiles >= 7 ? 4LL * teams : 0;
        if (const char *e = tune("DFX_HALF_UNITS")) nh = std::max(0, atoi(e));
        nh = std::min<long long>(nh, (long long)h->geom.total_units - (long long)(h->geom.static_rounds + 1) * teams);
        if (nh > 0) {
          h->geom.half_from = h->geom.total_units - (int)nh;
          h->geom.total_units += (int)nh;
          h->geom.claim_limit =
              (int)std::min<long long>(0x7ffffff0LL, (long long)h->geom.ntu * ((long long)h->geom.total_units + 4));
        }
      }
    }
#ifdef DFX_STAMPS
    // [grid][16 waves][16] sums, then [grid][16 waves][8 events][4] timeline words
    if (hipMalloc((void **)&h->d_prof, (size_t)h->grid * 768 * 8) != hipSuccess ||
        hipMemset(h->d_prof, 0, (size_t)h->grid * 768 * 8) != hipSuccess) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot allocate the stamp buffer");
    }
    h->geom.prof = h->d_prof;
#endif
#ifdef DFX_TRACE
    {
      void *hp = nullptr, *dp = nullptr;
      if (hipHostMalloc(&hp, (size_t)h->grid * 16 * 4 * 4, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
        conv_release(h);
        return fail(DFX_ERR_HIP, "conv_create: cannot allocate the trace buffer");
      }
      memset(hp, 0, (size_t)h->grid * 16 * 4 * 4);
      h->trace_host = (int *)hp;
      h->geom.trace = (int *)dp;
    }
#endif
    a.rows_per_unit = h->geom.th;
    a.units_per_image = h->geom.uy * h->geom.ux;
    snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_mfma_fused_kernel<%d,%d,%d,%d%s>", h->icb,
             h->ocb, h->G, d.dst_dt, fused ? "" : ",unfused");
    if (mfma_dispatch(h, h->args, h->geom, nullptr, 1) != 0) {
      conv_release(h);
      return fail(DFX_ERR_HIP, "conv_create: cannot raise dynamic LDS limit to %d bytes", h->lds);
    }
  } else {
    h->variant = DFX_VARIANT_GENERIC;
    h->block = 256;
    launch_conv_generic(a, nullptr, &h->grid, &h->lds);
    a.rows_per_unit = 0;
    // The streamed- and direct-weight MFMA kernels hold dst offsets in 32 bits: an op whose dst reaches 4 GiB
    // runs on the scalar kernel, and its name says why (dfx_conv_query).
    const bool big_dst = (long long)d.bs * d.oh * d.ow * (d.oc1x1 ? d.oc1x1 : d.oc) * (long long)dt_size(d.dst_dt) >= (1LL << 32) - 16;
    snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_generic_kernel<dt=%d>%s", d.dst_dt,
             big_dst && d.force_variant != DFX_VARIANT_GENERIC ? " [dst >= 4 GiB: no MFMA kernel (32-bit dst offsets)]" : "");
  }
  *out = h;
  return DFX_OK;
}

static float bias_to_f32(const void *b, int dt, int c) {
  switch (dt) {  // vcvtdq2ps after vpmovsxbd / vpmovzxbd / vmovups, jit_conv_kernel.cc:235-255
    case DFX_F32: return ((const float *)b)[c];
    case DFX_S32: return (float)((const int32_t *)b)[c];
    case DFX_S8: return (float)((const int8_t *)b)[c];
    case DFX_U8: return (float)((const uint8_t *)b)[c];
  }
  return 0.0f;
}

// One channel's precondition of the fast requant path (conv_mfma.cuh store_group<FAST>):
// amax bounds |true accumulator|; comp + bias must fold into ONE exact f32 add (bias
// integer-valued, |comp + bias| < 2^24, |acc + bias| < 2^24) and nothing may reach +-2^31.
static bool fast_ok_channel(double amax, double comp, float bias, float scale) {
  if (!std::isfinite(bias) || !std::isfinite(scale)) return false;
  const double b = bias;
  if (b != std::floor(b)) return false;
  const double lim = 16777216.0;  // 2^24
  if (std::fabs(comp + b) >= lim || amax + std::fabs(b) >= lim || std::fabs(comp) + amax >= lim) return false;
  return (amax + std::fabs(b)) * std::fabs((double)scale) * 1.0001 + 2.0 < 2147480000.0;
}

// Packs weights for conv_stream.cuh: ONE device buffer
//   [conv0 steps | conv1 steps | consts], a step = 2 k-blocks x (OCC | G) fragments of 1 KB,
// in the exact order the kernel walks them (oc chunk, ic chunk, step; 1x1 group, step).
static int set_weights_direct(dfx_conv_t *h, const int8_t *wei, const void *bia0, const float *scales0,
                              const int8_t *wei1, const void *bia1, const float *scales1);

static int set_weights_stream(dfx_conv_t *h, const int8_t *wei, const void *bia0, const float *scales0,
                              const int8_t *wei1, const void *bia1, const float *scales1) {
  if (h->direct) return set_weights_direct(h, wei, bia0, scales0, wei1, bia1, scales1);
  const dfx_conv_desc &d = h->d;
  const StreamGeom &g = h->sgeom;
  const bool fused = d.oc1x1 > 0;
  const int OCC = h->occ, G = h->G, OC = d.oc, IC = d.ic, OC1 = d.oc1x1;
  const int OCP = 32 * g.ocb, OC1P = fused ? 32 * G * g.n_g1 : 0;
  const size_t n0 = (size_t)g.s0_steps * 2 * OCC * 1024;
  const size_t n1 = fused ? (size_t)g.n_g1 * g.ks2 * 2 * G * 1024 : 0;
  std::vector<int8_t> pk(n0 + n1, 0);
  const int ntap = d.kh * d.kw;
  size_t o = 0;
  for (int occ = 0; occ < g.n_occ; ++occ)
    for (int icc = 0; icc < g.n_icc; ++icc) {
      const int kbn = std::min(2, g.icb - 2 * icc), ns = ntap * kbn;
      for (int s2 = 0; s2 < (ns + 1) / 2; ++s2)
        for (int j = 0; j < 2; ++j)
          for (int r = 0; r < OCC; ++r)
            for (int lane = 0; lane < 64; ++lane)
              for (int b = 0; b < 16; ++b, ++o) {
                const int s = 2 * s2 + j;
                if (s >= ns) continue;  // padding k-block: zero weights
                const int tap = s / kbn, icbl = s % kbn;
                // fused: A operand, row = channel 32*(occ*OCC + r) + rho.  unfused: B operand with
                // the store stage's channel permutation: column lam of block r = 32*OCC*occ + OCC*lam + r
                const int oc = fused ? 32 * (occ * OCC + r) + (lane & 31) : 32 * OCC * occ + OCC * (lane & 31) + r;
                const int ic = 64 * icc + 32 * icbl + 16 * (lane >> 5) + b;
                if (oc < OC && ic < IC) pk[o] = wei[dfx_blocked_offset(oc, ic, tap / d.kw, tap % d.kw, IC, d.kh, d.kw)];
              }
    }
  if (o != n0) return fail(DFX_ERR_HIP, "internal: conv0 pack size mismatch");
  for (int g1 = 0; g1 < g.n_g1; ++g1)
    for (int s2 = 0; s2 < g.ks2; ++s2)
      for (int j = 0; j < 2; ++j)
        for (int cc = 0; cc < G; ++cc)
          for (int lane = 0; lane < 64; ++lane)
            for (int b = 0; b < 16; ++b, ++o) {
              const int blk = 2 * s2 + j;
              if (blk >= g.ocb) continue;
              const int oc1 = 32 * G * g1 + G * (lane & 31) + cc;
              const int oc = 32 * blk + 8 * (b >> 2) + 4 * (lane >> 5) + (b & 3);  // mid's k order
              if (oc1 < OC1 && oc < OC) pk[o] = wei1[dfx_blocked_offset(oc1, oc, 0, 0, OC, 1, 1)];
            }
  // consts: comp0 (s32) bias0 scale0 [OCP each] comp1 (s32) bias1 scale1 [OC1P each]; padding
  // channels are all-zero (their intermediate is 0 and meets zero 1x1 weights)
  std::vector<int32_t> cst((size_t)3 * (OCP + OC1P), 0);
  auto put_f = [&](size_t idx, float v) { memcpy(&cst[idx], &v, 4); };
  bool fast = d.conv0_round_mode == DFX_ROUND_NEAREST && (!fused || d.conv1_round_mode == DFX_ROUND_NEAREST);
  std::vector<float> fb0(OC), fb1(OC1 ? OC1 : 1);  // bias as f32
  for (int c = 0; c < OC; ++c) {
    int32_t sum = 0, pos = 0, neg = 0;
    for (int ic = 0; ic < IC; ++ic)
      for (int tap = 0; tap < ntap; ++tap) {
        const int w = wei[dfx_blocked_offset(c, ic, tap / d.kw, tap % d.kw, IC, d.kh, d.kw)];
        sum += w;
        (w > 0 ? pos : neg) += w;
      }
    cst[c] = 128 * sum;
    fb0[c] = d.bia0_dt == DFX_UNDEF ? 0.0f : bias_to_f32(bia0, d.bia0_dt, c);
    const float sc = scales0[d.conv0_nscales > 1 ? c : 0];
    put_f((size_t)OCP + c, fb0[c]);
    put_f((size_t)2 * OCP + c, sc);
    fast = fast && fast_ok_channel(255.0 * std::max(pos, -neg), 128.0 * sum, fb0[c], sc);
  }
  for (int c = 0; c < OC1; ++c) {
    int32_t sum = 0, pos = 0, neg = 0;
    for (int oc = 0; oc < OC; ++oc) {
      const int w = wei1[dfx_blocked_offset(c, oc, 0, 0, OC, 1, 1)];
      sum += w;
      (w > 0 ? pos : neg) += w;
    }
    cst[(size_t)3 * OCP + c] = 128 * sum;
    fb1[c] = d.bia1_dt == DFX_UNDEF ? 0.0f : bias_to_f32(bia1, d.bia1_dt, c);
    const float sc = scales1[d.conv1_nscales > 1 ? c : 0];
    put_f((size_t)3 * OCP + OC1P + c, fb1[c]);
    put_f((size_t)3 * OCP + 2 * OC1P + c, sc);
    fast = fast && fast_ok_channel(255.0 * std::max(pos, -neg), 128.0 * sum, fb1[c], sc);
  }
  if (const char *e = tune("DFX_NO_FAST")) fast = fast && atoi(e) == 0;  // testing aid: force the exact path
  if (fast) {  // the fast path reads comp + bias (an exact f32) from the bias slot
    for (int c = 0; c < OC; ++c) put_f((size_t)OCP + c, (float)((double)cst[c] + (double)fb0[c]));
    for (int c = 0; c < OC1; ++c)
      put_f((size_t)3 * OCP + OC1P + c, (float)((double)cst[(size_t)3 * OCP + c] + (double)fb1[c]));
  }
  h->sgeom.fast = fast ? 1 : 0;
  if (!h->d_wei) HIP_TRY(hipMalloc(&h->d_wei, pk.size() + cst.size() * 4));
  char *base = (char *)h->d_wei;
  HIP_TRY(hipMemcpy(base, pk.data(), pk.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(base + pk.size(), cst.data(), cst.size() * 4, hipMemcpyHostToDevice));
  h->args.wei = (const int8_t *)base;
  h->args.wei1 = (const int8_t *)(base + n0);
  h->args.consts = (const float *)(base + pk.size());
  h->weights_set = true;
  return DFX_OK;
}

// Packs weights for conv_direct.cuh: ONE device buffer [W0d | W1d | consts].
//   W0d[ob][kb = icb * ntap + tap][lane][16]: byte b of lane = W0[oc = 32 ob + (lane & 31)][ic = 32 icb + 16 (lane >> 5) + b][tap]
//   W1d[g1][blk][cc][lane][16]: byte b = W1[oc1 = 32 G g1 + G (lane & 31) + cc][oc = 32 blk + 8 (b >> 2) + 4 (lane >> 5) + (b & 3)]
static int set_weights_direct(dfx_conv_t *h, const int8_t *wei, const void *bia0, const float *scales0,
                              const int8_t *wei1, const void *bia1, const float *scales1) {
  const dfx_conv_desc &d = h->d;
  const DirectGeom &g = h->dgeom;
  const int G = h->G, OC = d.oc, IC = d.ic, OC1 = d.oc1x1;
  const int OCP = 32 * g.ocb, OC1P = 32 * G * g.n_g1;
  const int ntap = d.kh * d.kw, nkb0 = g.icb * ntap;
  const size_t n0 = (size_t)g.ocb * nkb0 * 1024, n1 = (size_t)g.n_g1 * g.ocb * G * 1024;
  std::vector<int8_t> pk(n0 + n1, 0);
  size_t o = 0;
  for (int ob = 0; ob < g.ocb; ++ob)
    for (int icb = 0; icb < g.icb; ++icb)
      for (int tap = 0; tap < ntap; ++tap)
        for (int lane = 0; lane < 64; ++lane)
          for (int b = 0; b < 16; ++b, ++o) {
            const int oc = 32 * ob + (lane & 31), ic = 32 * icb + 16 * (lane >> 5) + b;
            if (oc < OC && ic < IC) pk[o] = wei[dfx_blocked_offset(oc, ic, tap / d.kw, tap % d.kw, IC, d.kh, d.kw)];
          }
  for (int g1 = 0; g1 < g.n_g1; ++g1)
    for (int blk = 0; blk < g.ocb; ++blk)
      for (int cc = 0; cc < G; ++cc)
        for (int lane = 0; lane < 64; ++lane)
          for (int b = 0; b < 16; ++b, ++o) {
            const int oc1 = 32 * G * g1 + G * (lane & 31) + cc;
            const int oc = 32 * blk + 8 * (b >> 2) + 4 * (lane >> 5) + (b & 3);
            if (oc1 < OC1 && oc < OC) pk[o] = wei1[dfx_blocked_offset(oc1, oc, 0, 0, OC, 1, 1)];
          }
  if (o != n0 + n1) return fail(DFX_ERR_HIP, "internal: direct pack size mismatch");
  std::vector<int32_t> cst((size_t)3 * (OCP + OC1P), 0);
  auto put_f = [&](size_t idx, float v) { memcpy(&cst[idx], &v, 4); };
  bool fast = d.conv0_round_mode == DFX_ROUND_NEAREST && (OC1 == 0 || d.conv1_round_mode == DFX_ROUND_NEAREST);
  std::vector<float> fb0(OC), fb1(OC1 ? OC1 : 1);
  for (int c = 0; c < OC; ++c) {
    int32_t sum = 0, pos = 0, neg = 0;
    for (int ic = 0; ic < IC; ++ic)
      for (int tap = 0; tap < ntap; ++tap) {
        const int w = wei[dfx_blocked_offset(c, ic, tap / d.kw, tap % d.kw, IC, d.kh, d.kw)];
        sum += w;
        (w > 0 ? pos : neg) += w;
      }
    cst[c] = 128 * sum;
    fb0[c] = d.bia0_dt == DFX_UNDEF ? 0.0f : bias_to_f32(bia0, d.bia0_dt, c);
    const float sc = scales0[d.conv0_nscales > 1 ? c : 0];
    put_f((size_t)OCP + c, fb0[c]);
    put_f((size_t)2 * OCP + c, sc);
    fast = fast && fast_ok_channel(255.0 * std::max(pos, -neg), 128.0 * sum, fb0[c], sc);
  }
  for (int c = 0; c < OC1; ++c) {
    int32_t sum = 0, pos = 0, neg = 0;
    for (int oc = 0; oc < OC; ++oc) {
      const int w = wei1[dfx_blocked_offset(c, oc, 0, 0, OC, 1, 1)];
      sum += w;
      (w > 0 ? pos : neg) += w;
    }
    cst[(size_t)3 * OCP + c] = 128 * sum;
    fb1[c] = d.bia1_dt == DFX_UNDEF ? 0.0f : bias_to_f32(bia1, d.bia1_dt, c);
    const float sc = scales1[d.conv1_nscales > 1 ? c : 0];
    put_f((size_t)3 * OCP + OC1P + c, fb1[c]);
    put_f((size_t)3 * OCP + 2 * OC1P + c, sc);
    fast = fast && fast_ok_channel(255.0 * std::max(pos, -neg), 128.0 * sum, fb1[c], sc);
  }
  if (const char *e = tune("DFX_NO_FAST")) fast = fast && atoi(e) == 0;  // testing aid: force the exact path
  if (fast) {  // the fast path reads comp + bias (an exact f32) from the bias slot
    for (int c = 0; c < OC; ++c) put_f((size_t)OCP + c, (float)((double)cst[c] + (double)fb0[c]));
    for (int c = 0; c < OC1; ++c)
      put_f((size_t)3 * OCP + OC1P + c, (float)((double)cst[(size_t)3 * OCP + c] + (double)fb1[c]));
  }
  h->dgeom.fast = fast ? 1 : 0;
  // Requant without int -> float conversions (conv_direct.cuh, round 3), proven per channel from the actual
  // weights like the modes of the resident kernels: with activations stored as u8 - 128 the raw accumulator lies
  // in [-(128 P + 127 N), 127 P + 128 N] (P / N = sums of the positive / negative weights' magnitudes).
  //   m0  stage 0 "fma": start value bits(2^23) + comp + bias, t = acc + bias inside (-2^23, 2^23), bias
  //       integer-valued, scale >= 0: comp slot <- start bits, bias slot <- -2^23 * scale
  //   m1  stage 1 "magic" (u8 output through the 16-byte store path only): start 1/(2 pi), conditions of
  //       magic1_ok; bias slot <- (comp + bias - 2^23 - 0x22F983) * 2^-26, scale slot <- scale * 2^26 (m1 = 2),
  //       or bias slot <- (comp + bias - 2^23 - 0x22F983) * scale where that product is exact for every channel:
  //       one fma (m1 = 3)
  h->dgeom.m0 = h->dgeom.m1 = 0;
  const bool no_magic = tune("DFX_NO_MAGIC") && atoi(tune("DFX_NO_MAGIC")) != 0;
  if (fast && !no_magic) {
    auto pn = [&](bool stage1, int c, double &P, double &N) {
      P = N = 0;
      if (!stage1) {
        for (int ic = 0; ic < IC; ++ic)
          for (int tap = 0; tap < ntap; ++tap) {
            const int w = wei[dfx_blocked_offset(c, ic, tap / d.kw, tap % d.kw, IC, d.kh, d.kw)];
            (w > 0 ? P : N) += std::abs(w);
          }
      } else {
        for (int oc = 0; oc < OC; ++oc) {
          const int w = wei1[dfx_blocked_offset(c, oc, 0, 0, OC, 1, 1)];
          (w > 0 ? P : N) += std::abs(w);
        }
      }
    };
    // (stage 0's "fma" route relies on the unsigned saturation of a u8 result: the fused intermediate, or a u8 dst)
    bool m0 = OC1 > 0 || d.dst_dt == DFX_U8;
    for (int c = 0; c < OC && m0; ++c) {
      double P, N;
      pn(false, c, P, N);
      const float sc = scales0[d.conv0_nscales > 1 ? c : 0];
      const double b = fb0[c], cb = 128.0 * (P - N) + b;
      m0 = b == std::floor(b) && sc >= 0.0f && std::isfinite(sc * 8388608.0f) &&
           -(128.0 * P + 127.0 * N) + cb > -8388608.0 && 127.0 * P + 128.0 * N + cb < 8388608.0;
    }
    if (m0) {
      for (int c = 0; c < OC; ++c) {
        const float sc = scales0[d.conv0_nscales > 1 ? c : 0];
        cst[c] = MAGIC3_BITS + cst[c] + (int32_t)fb0[c];
        put_f((size_t)OCP + c, -8388608.0f * sc);
      }
      h->dgeom.m0 = 1;
    }
    bool m1 = d.dst_dt == DFX_U8 && G == 4 && OC1 > 0, fma1 = true;
    for (int c = 0; c < OC1 && m1; ++c) {
      double P, N;
      pn(true, c, P, N);
      const float sc = scales1[d.conv1_nscales > 1 ? c : 0];
      const double b = fb1[c], cb = 128.0 * (P - N) + b, lo = -(128.0 * P + 127.0 * N), hi = 127.0 * P + 128.0 * N;
      const double k = cb - 8388608.0 - (double)0x22F983;
      m1 = b == std::floor(b) && lo >= (double)MAGIC1_LO && hi <= (double)MAGIC1_HI && std::fabs(k) < 16777216.0 &&
           std::fabs(lo + cb) < 16777216.0 && std::fabs(hi + cb) < 16777216.0 && std::isfinite(sc * 67108864.0f);
      const double prod = k * (double)sc;
      fma1 = fma1 && std::isfinite(prod) && (double)(float)prod == prod;
    }
    m1 = m1 && m0;  // (the kernels specialise on "both stages without conversions": conv_direct.cuh, QM)
    if (m1) {
      for (int c = 0; c < OC1; ++c) {
        const float sc = scales1[d.conv1_nscales > 1 ? c : 0];
        const double k = (double)cst[(size_t)3 * OCP + c] + (double)fb1[c] - 8388608.0 - (double)0x22F983;
        put_f((size_t)3 * OCP + OC1P + c, fma1 ? (float)(k * (double)sc) : (float)k * 1.4901161193847656e-08f);
        put_f((size_t)3 * OCP + 2 * OC1P + c, sc * 67108864.0f);
      }
      h->dgeom.m1 = fma1 ? 3 : 2;
    }
  }
#ifdef DK_DEBUG
  h->dgeom.wei_bytes = (long long)n0; h->dgeom.wei1_bytes = (long long)n1; h->dgeom.cst_bytes = (long long)cst.size() * 4;
#endif
  if (!h->d_wei) HIP_TRY(hipMalloc(&h->d_wei, pk.size() + cst.size() * 4));
  char *base = (char *)h->d_wei;
  HIP_TRY(hipMemcpy(base, pk.data(), pk.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(base + pk.size(), cst.data(), cst.size() * 4, hipMemcpyHostToDevice));
  h->args.wei = (const int8_t *)base;
  h->args.wei1 = (const int8_t *)(base + n0);
  h->args.consts = (const float *)(base + pk.size());
  h->weights_set = true;
  if (tune("DFX_DEBUG_PTRS")) fprintf(stderr, "[dfx] direct weights %p..%p (W0 %zu, W1 %zu, consts %zu bytes)\n", (void *)base, (void *)(base + pk.size() + cst.size() * 4), n0, n1, cst.size() * 4);
  return DFX_OK;
}

int dfx_conv_set_weights(dfx_conv_t *h, const int8_t *wei, const void *bia0, const float *scales0,
                         const int8_t *wei1, const void *bia1, const float *scales1) {
  if (!h || !wei || !scales0) return fail(DFX_ERR_INVALID, "set_weights: null argument");
  DeviceGuard dg(h->device);
  const dfx_conv_desc &d = h->d;
  const bool fused = d.oc1x1 > 0;
  if (fused && (!wei1 || !scales1)) return fail(DFX_ERR_INVALID, "set_weights: fused op needs wei1x1 and scales1");
  if ((d.bia0_dt != DFX_UNDEF && !bia0) || (fused && d.bia1_dt != DFX_UNDEF && !bia1))
    return fail(DFX_ERR_INVALID, "set_weights: bias dtype set but pointer is null");

  const int OC = d.oc, OC1 = d.oc1x1, IC = d.ic;
  const size_t nw0 = (size_t)OC * IC * d.kh * d.kw, nw1 = (size_t)OC1 * OC;
  std::vector<int8_t> p0(nw0), p1(nw1 ? nw1 : 1);
  std::vector<float> cst((size_t)3 * (OC + OC1), 0.0f);
  float *comp0 = cst.data();  // f32: exact, |comp| < 2^24 for K <= 1023
  float *b0 = cst.data() + OC, *s0 = cst.data() + 2 * OC;
  float *comp1 = cst.data() + 3 * OC;
  float *b1 = cst.data() + 3 * OC + OC1, *s1 = cst.data() + 3 * OC + 2 * OC1;
  for (int c = 0; c < OC; ++c) {
    b0[c] = d.bia0_dt == DFX_UNDEF ? 0.0f : bias_to_f32(bia0, d.bia0_dt, c);
    s0[c] = scales0[d.conv0_nscales > 1 ? c : 0];  // count 1 = broadcast (intended semantics)
  }
  for (int c = 0; c < OC1; ++c) {
    b1[c] = d.bia1_dt == DFX_UNDEF ? 0.0f : bias_to_f32(bia1, d.bia1_dt, c);
    s1[c] = scales1[d.conv1_nscales > 1 ? c : 0];
  }

  if (h->variant == DFX_VARIANT_MFMA_STREAM) return set_weights_stream(h, wei, bia0, scales0, wei1, bia1, scales1);

  if (h->variant != DFX_VARIANT_GENERIC) {
    const int ICB = h->icb, OCB = h->ocb, G = h->G, NCB = OC1 / 32;
    // W0 fragments [r][tap][c][lane][16]: lane (rho = lane&31, hh = lane>>5), byte j
    //   = W0[oc = 32r + rho][ic = 32c + 16hh + j][tap]
    for (int r = 0; r < OCB; ++r)
      for (int tap = 0; tap < 9; ++tap)
        for (int c = 0; c < ICB; ++c)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
              // fused: A operand, row = oc 32r + rho.  unfused: B operand with the channel
              // permutation of the store stage (G == OCB): column lam of block r = oc G*lam + r
              const int oc = fused ? 32 * r + (lane & 31) : G * (lane & 31) + r;
              const int ic = 32 * c + 16 * (lane >> 5) + j;
              p0[((((size_t)r * 9 + tap) * ICB + c) * 64 + lane) * 16 + j] =
                  wei[dfx_blocked_offset(oc, ic, tap / 3, tap % 3, IC, 3, 3)];
            }
    // W1 fragments [cb][r][lane][16]: cb = cg*G + cc; lane (lam, hh); byte j = 4q + i
    //   = W1[oc1 = 32G*cg + G*lam + cc][oc = 32r + 8q + 4hh + i]
    for (int cb = 0; cb < NCB; ++cb)
      for (int r = 0; r < OCB; ++r)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 16; ++j) {
            const int cg = cb / G, cc = cb % G, lam = lane & 31, hh = lane >> 5;
            const int oc1 = 32 * G * cg + G * lam + cc;
            const int oc = 32 * r + 8 * (j >> 2) + 4 * hh + (j & 3);
            p1[(((size_t)cb * OCB + r) * 64 + lane) * 16 + j] =
                wei1[dfx_blocked_offset(oc1, oc, 0, 0, OC, 1, 1)];
          }
    // ---- constants and requant mode per stage (conv_mfma.cuh header; emit_group) ----
    // Per channel, from the ACTUAL weights: P = sum of positive weights, N = -sum of negative
    // ones.  With activations stored as u8 - 128 in [-128, 127] the raw MFMA accumulator lies in
    // [-(128 P + 127 N), 127 P + 128 N]; comp = 128 * (P - N) turns it into the reference's s32
    // accumulator, |true acc| <= 255 * max(P, N).
    struct Ch { double P, N; };
    std::vector<Ch> c0(OC), c1(OC1 ? OC1 : 1);
    for (int oc = 0; oc < OC; ++oc) {
      double P = 0, N = 0;
      for (int ic = 0; ic < IC; ++ic)
        for (int tap = 0; tap < 9; ++tap) {
          const int w = wei[dfx_blocked_offset(oc, ic, tap / 3, tap % 3, IC, 3, 3)];
          (w > 0 ? P : N) += std::abs(w);
        }
      c0[oc] = {P, N};
    }
    for (int o1 = 0; o1 < OC1; ++o1) {
      double P = 0, N = 0;
      for (int oc = 0; oc < OC; ++oc) {
        const int w = wei1[dfx_blocked_offset(o1, oc, 0, 0, OC, 1, 1)];
        (w > 0 ? P : N) += std::abs(w);
      }
      c1[o1] = {P, N};
    }
    // "fast": rounding to nearest-even (the caller checks the round mode), every value finite and
    // |f| < 2^31 so the x86 overflow / NaN results of vcvtps2dq cannot occur
    auto fast_ok = [](const Ch &c, float bias, float scale) {
      if (!std::isfinite(bias) || !std::isfinite(scale)) return false;
      const double amax = 255.0 * std::max(c.P, c.N);
      return (amax + std::fabs((double)bias)) * std::fabs((double)scale) * 1.0001 + 2.0 < 2147480000.0;
    };
    // "magic", accumulator started from bits(1.5 * 2^23) + comp + bias (stage 0 of a fused op):
    // bias integer-valued and raw + comp + bias inside the binade, i.e. |.| < 2^22
    auto magic0_ok = [](const Ch &c, float bias) {
      const double b = bias, cb = 128.0 * (c.P - c.N) + b;
      if (b != std::floor(b)) return false;
      return -(128.0 * c.P + 127.0 * c.N) + cb > -4194304.0 && 127.0 * c.P + 128.0 * c.N + cb < 4194304.0;
    };
    // "magic", accumulator started from the inline constant 1/(2 pi) = 0x3E22F983 (ulp 2^-26):
    // raw within the mantissa's room; k = (comp + bias - 2^23 - 0x22F983) exactly representable;
    // |acc + bias| < 2^24; scale * 2^26 finite
    auto magic1_ok = [](const Ch &c, float bias, float scale) {
      const double b = bias, cb = 128.0 * (c.P - c.N) + b;
      if (b != std::floor(b)) return false;
      const double lo = -(128.0 * c.P + 127.0 * c.N), hi = 127.0 * c.P + 128.0 * c.N;
      if (lo < (double)MAGIC1_LO || hi > (double)MAGIC1_HI) return false;
      if (std::fabs(cb - 8388608.0 - (double)0x22F983) >= 16777216.0) return false;
      if (std::fabs(lo + cb) >= 16777216.0 || std::fabs(hi + cb) >= 16777216.0) return false;
      return std::isfinite(scale * 67108864.0f);
    };
    const bool no_fast = tune("DFX_NO_FAST") && atoi(tune("DFX_NO_FAST")) != 0;     // testing aid: exact paths
    const bool no_magic = tune("DFX_NO_MAGIC") && atoi(tune("DFX_NO_MAGIC")) != 0;  // testing aid: no magic paths
    int mode0 = (d.conv0_round_mode == DFX_ROUND_NEAREST && !no_fast) ? 2 : 0;
    for (int oc = 0; oc < OC && mode0; ++oc) {
      if (!fast_ok(c0[oc], b0[oc], s0[oc])) mode0 = 0;
      else if (mode0 == 2 && !(fused ? magic0_ok(c0[oc], b0[oc]) : magic1_ok(c0[oc], b0[oc], s0[oc]))) mode0 = 1;
    }
    int mode1 = (fused && d.conv1_round_mode == DFX_ROUND_NEAREST && !no_fast) ? 2 : 0;
    for (int o1 = 0; o1 < OC1 && mode1; ++o1) {
      if (!fast_ok(c1[o1], b1[o1], s1[o1])) mode1 = 0;
      else if (mode1 == 2 && !magic1_ok(c1[o1], b1[o1], s1[o1])) mode1 = 1;
    }
    if (no_magic) { mode0 = std::min(mode0, 1); mode1 = std::min(mode1, 1); }
    // "fma" (stage 0 of the role-specialised kernel, conv_mfma_roles.cuh): accumulator started from
    // bits(2^23) + comp + bias.  t = acc + bias must stay inside (-2^23, 2^23): non-negative t then reads as the
    // float 2^23 + t, negative t as 2^23 - |t|/2; fma(x, s, -2^23 s) = t * s with one rounding (2^23 s is exact)
    // or something negative that the stage's ReLU + u8 saturation turn into 0 like the reference's negative
    // product -- which needs s >= 0.  bias integer-valued; |acc| <= 2^24 follows (0 is a possible acc).
    auto fma0_ok = [](const Ch &c, float bias, float scale) {
      const double b = bias, cb = 128.0 * (c.P - c.N) + b;
      if (b != std::floor(b) || !(scale >= 0.0f) || !std::isfinite(scale * 8388608.0f)) return false;
      return -(128.0 * c.P + 127.0 * c.N) + cb > -8388608.0 && 127.0 * c.P + 128.0 * c.N + cb < 8388608.0;
    };
    bool roles = h->roles_ok && fused && mode1 == 2 && d.conv0_round_mode == DFX_ROUND_NEAREST && !no_fast && !no_magic;
    for (int oc = 0; oc < OC && roles; ++oc)
      if (!fast_ok(c0[oc], b0[oc], s0[oc]) || !fma0_ok(c0[oc], b0[oc], s0[oc])) roles = false;
    if (roles) mode0 = 3;
    h->roles = roles;
    // "fma" for stage 1 (role-specialised kernel only): (x + B) * C = fma(x, C, B * C) with one rounding when
    // the addend B * C = (comp + bias - 2^23 - 0x22F983) * scale is exactly representable -- power-of-two scales
    // and a few others; checked per channel in double (a 24-bit integer times a 24-bit mantissa is exact there)
    if (roles) {
      bool fma1 = true;
      for (int o1 = 0; o1 < OC1 && fma1; ++o1) {
        const double k = 128.0 * (c1[o1].P - c1[o1].N) + (double)b1[o1] - 8388608.0 - (double)0x22F983;
        const double prod = k * (double)s1[o1];
        fma1 = std::isfinite(prod) && (double)(float)prod == prod;
      }
      if (fma1) mode1 = 3;
    }
    if (h->variant == DFX_VARIANT_MFMA_FUSED) {
      // (the suffix names stage 1's requant route: "fma" = one v_fma_f32 per value, admitted where the addend is exactly
      //  representable -- power-of-two scales and a few others; "magic" = v_add_f32 + v_mul_f32)
      if (roles) snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_mfma_roles_kernel<%d,%d,%d,%d>/%s", ICB, OCB, NCB, d.dst_dt,
                          mode1 == 3 ? "fma" : "magic");
      else snprintf(h->kernel_name, sizeof(h->kernel_name), "conv_mfma_fused_kernel<%d,%d,%d,%d>", ICB, OCB, G, d.dst_dt);
      h->block = roles ? RL_THREADS : MFMA_THREADS;
    }
    // slot A is an integer (bit copy into the f32 array), B and C are floats
    auto put_i = [](float *dst, int32_t v) { memcpy(dst, &v, 4); };
    auto inline_stage = [&](int mode, const Ch &c, float bias, float scale, float *A, float *B, float *C) {
      const int32_t comp = (int32_t)(128.0 * (c.P - c.N));
      put_i(A, comp - MAGIC1_BITS);  // acc bits + A = the reference's s32 accumulator
      if (mode == 3) {  // fma(x, C, B): B = (comp + bias - 2^23 - 0x22F983) * scale (proven exact), C = scale * 2^26
        *B = (float)(((double)comp + (double)bias - 8388608.0 - (double)0x22F983) * (double)scale);
        *C = scale * 67108864.0f;
      } else if (mode == 2) {
        *B = (float)((double)comp + (double)bias - 8388608.0 - (double)0x22F983) * 1.4901161193847656e-08f;  // * 2^-26, exact
        *C = scale * 67108864.0f;                                                                           // * 2^26, exact
      } else {
        *B = bias;
        *C = scale;
      }
    };
    for (int oc = 0; oc < OC; ++oc) {
      if (fused) {
        const int32_t comp = (int32_t)(128.0 * (c0[oc].P - c0[oc].N));
        put_i(comp0 + oc, mode0 == 3 ? MAGIC3_BITS + comp + (int32_t)b0[oc]
                        : mode0 == 2 ? MAGIC0_BITS + comp + (int32_t)b0[oc] : comp);  // accumulator start value
        if (mode0 == 3) b0[oc] = -8388608.0f * s0[oc];  // the fma's addend -2^23 * scale (exact)
      } else {
        float A, B, C;
        inline_stage(mode0, c0[oc], b0[oc], s0[oc], &A, &B, &C);
        comp0[oc] = A; b0[oc] = B; s0[oc] = C;
      }
    }
    for (int o1 = 0; o1 < OC1; ++o1) {
      float A, B, C;
      inline_stage(mode1, c1[o1], b1[o1], s1[o1], &A, &B, &C);
      comp1[o1] = A; b1[o1] = B; s1[o1] = C;
    }
    h->geom.mode0 = mode0;
    h->geom.mode1 = mode1;
    h->geom.s0_uniform = (fused && d.conv0_nscales == 1) ? 1 : 0;
    h->geom.s0_value = scales0[0];
  } else {
    memcpy(p0.data(), wei, nw0);
    if (fused) memcpy(p1.data(), wei1, nw1);
  }

  if (h->variant != DFX_VARIANT_GENERIC) {
    // constants in the kernel's layout: the storing stage's B and C as pairs {k, k} (conv_mfma.cuh,
    // mfma_cst_floats)
    {
      std::vector<float> lay((size_t)mfma_cst_floats(OC, OC1), 0.0f);
      if (fused) {
        memcpy(lay.data(), cst.data(), (size_t)(3 * OC + OC1) * 4);  // A0 B0 C0 A1
        for (int c = 0; c < OC1; ++c) {
          lay[(size_t)3 * OC + OC1 + 2 * c] = lay[(size_t)3 * OC + OC1 + 2 * c + 1] = b1[c];
          lay[(size_t)3 * OC + 3 * OC1 + 2 * c] = lay[(size_t)3 * OC + 3 * OC1 + 2 * c + 1] = s1[c];
        }
      } else {
        memcpy(lay.data(), cst.data(), (size_t)OC * 4);  // A0
        for (int c = 0; c < OC; ++c) {
          lay[(size_t)OC + 2 * c] = lay[(size_t)OC + 2 * c + 1] = b0[c];
          lay[(size_t)3 * OC + 2 * c] = lay[(size_t)3 * OC + 2 * c + 1] = s0[c];
        }
      }
      cst.swap(lay);
    }
    // one buffer in LDS-image order: [W0 fragments | W1 fragments | constants]
    const size_t cbytes = round16(cst.size() * 4);
    if (!h->d_wei) HIP_TRY(hipMalloc(&h->d_wei, nw0 + nw1 + cbytes));
    char *base = (char *)h->d_wei;
    HIP_TRY(hipMemcpy(base, p0.data(), nw0, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + nw0, p1.data(), nw1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(base + nw0 + nw1, cst.data(), cst.size() * 4, hipMemcpyHostToDevice));
    h->args.wei = (const int8_t *)base;
    h->args.wei1 = (const int8_t *)(base + nw0);
    h->args.consts = (const float *)(base + nw0 + nw1);
  } else {
    if (!h->d_wei) {
      HIP_TRY(hipMalloc(&h->d_wei, nw0));
      HIP_TRY(hipMalloc(&h->d_wei1, nw1 ? nw1 : 16));
      HIP_TRY(hipMalloc(&h->d_consts, cst.size() * 4));
    }
    HIP_TRY(hipMemcpy(h->d_wei, p0.data(), nw0, hipMemcpyHostToDevice));
    if (nw1) HIP_TRY(hipMemcpy(h->d_wei1, p1.data(), nw1, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_consts, cst.data(), cst.size() * 4, hipMemcpyHostToDevice));
    h->args.wei = (const int8_t *)h->d_wei;
    h->args.wei1 = (const int8_t *)h->d_wei1;
    h->args.consts = (const float *)h->d_consts;
  }
  h->weights_set = true;
  return DFX_OK;
}

int dfx_conv_submit(dfx_conv_t *h, const void *src_dev, void *dst_dev, dfx_stream_t s) {
  if (!h || !src_dev || !dst_dev) return fail(DFX_ERR_INVALID, "conv_submit: null argument");
  if (!h->weights_set) return fail(DFX_ERR_STATE, "conv_submit: dfx_conv_set_weights not called");
  DeviceGuard dg(h->device);
  // per-launch copies: concurrent submits of one handle (other host threads, other streams) share
  // only immutable state and each takes its own unit-queue slot of the ring
  ConvArgs a = h->args;
  a.src = (const uint8_t *)src_dev;
  a.dst = dst_dev;
  MfmaGeom g = h->geom;
  if (g.queue) {
    // one queue-ring slot per launch in flight; see struct dfx_conv for the guard
    const hipStream_t st = (hipStream_t)s;
    std::lock_guard<std::mutex> lk(*h->ring_mu);
    const unsigned slot = h->launch_seq++ % DFX_QUEUE_RING;
    if (h->launch_seq == 1) h->first_stream = st;
    if (!h->multi_stream && st != h->first_stream) {
      // second stream seen: the launches so far (all on first_stream, no events) are covered by ONE event
      // recorded there now -- a stream event stands for everything submitted before it
      h->multi_stream = true;
      hipEvent_t e0 = nullptr;
      HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
      hipError_t r = hipEventRecord(e0, h->first_stream);
      if (r == hipSuccess) r = hipStreamWaitEvent(st, e0, 0);
      (void)hipEventDestroy(e0);
      if (r != hipSuccess) HIP_TRY(hipDeviceSynchronize());  // (first_stream already destroyed by the caller)
      for (unsigned i = 0; i < DFX_QUEUE_RING; ++i)
        if (h->slot_state[i] == 1) h->slot_state[i] = 0;  // their launches are ordered before this one now
    }
    if (h->slot_state[slot] == 2 && h->slot_stream[slot] != st) HIP_TRY(hipStreamWaitEvent(st, h->slot_ev[slot], 0));
    g.queue += 2 * slot;
    if (mfma_dispatch(h, a, g, st, 0) != 0) return fail(DFX_ERR_UNSUPPORTED, "conv_submit: no kernel instance for this op");
    HIP_TRY(hipGetLastError());
    h->slot_stream[slot] = st;
    h->slot_state[slot] = 1;
    if (h->multi_stream) {
      if (!h->slot_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&h->slot_ev[slot], hipEventDisableTiming));
      HIP_TRY(hipEventRecord(h->slot_ev[slot], st));
      h->slot_state[slot] = 2;
    }
    return DFX_OK;
  }
  int rc;
  if (h->variant != DFX_VARIANT_GENERIC)
    rc = mfma_dispatch(h, a, g, (hipStream_t)s, 0);
  else
    rc = launch_conv_generic(a, (hipStream_t)s, nullptr, nullptr);
  if (rc != 0) return fail(DFX_ERR_UNSUPPORTED, "conv_submit: no kernel instance for this op");
  HIP_TRY(hipGetLastError());
  return DFX_OK;
}

static size_t conv_src_bytes(const dfx_conv_desc &d) { return (size_t)d.bs * d.ih * d.iw * d.ic; }
static size_t conv_dst_bytes(const dfx_conv_desc &d) {
  const size_t px = d.fuse_pool ? (size_t)(d.oh / 2) * (d.ow / 2) : (size_t)d.oh * d.ow;
  return (size_t)d.bs * px * (d.oc1x1 ? d.oc1x1 : d.oc) * dt_size(d.dst_dt);
}

int dfx_conv_submit_host(dfx_conv_t *h, const void *src_host, void *dst_host) {
  if (!h || !src_host || !dst_host) return fail(DFX_ERR_INVALID, "conv_submit_host: null argument");
  DeviceGuard dg(h->device);
  const size_t sb = conv_src_bytes(h->d), db = conv_dst_bytes(h->d);
  if (!h->d_src) {
    HIP_TRY(hipMalloc(&h->d_src, sb));
    HIP_TRY(hipMalloc(&h->d_dst, db));
    HIP_TRY(hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking));
  }
  HIP_TRY(hipMemcpyAsync(h->d_src, src_host, sb, hipMemcpyHostToDevice, h->host_stream));
  int rc = dfx_conv_submit(h, h->d_src, h->d_dst, h->host_stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(dst_host, h->d_dst, db, hipMemcpyDeviceToHost, h->host_stream));
  HIP_TRY(hipStreamSynchronize(h->host_stream));
  return DFX_OK;
}

int dfx_conv_query(const dfx_conv_t *h, dfx_conv_info *info) {
  if (!h || !info) return fail(DFX_ERR_INVALID, "conv_query: null argument");
  const dfx_conv_desc &d = h->d;
  memset(info, 0, sizeof(*info));
  info->variant = h->variant;
  info->grid = h->grid; info->block = h->block; info->lds_bytes = h->lds;
  info->rows_per_unit = h->args.rows_per_unit;
  info->device = h->device;
  const uint64_t px = (uint64_t)d.bs * d.oh * d.ow;
  const uint64_t mac = px * ((uint64_t)d.oc * d.ic * d.kh * d.kw + (uint64_t)d.oc1x1 * d.oc);
  info->algorithmic_ops = 2 * mac;
  info->algorithmic_bytes = conv_src_bytes(d) + conv_dst_bytes(d) +
                            (uint64_t)d.oc * d.ic * d.kh * d.kw + (uint64_t)d.oc1x1 * d.oc;
  snprintf(info->kernel_name, sizeof(info->kernel_name), "%s", h->kernel_name);
  return DFX_OK;
}

#ifdef DK_DEBUG
// bounds-checking diagnostic build of conv_direct.cuh: 32 x {offset, size} of the first violation per tag (-1 = none)
int dfx_debug_read_bounds(dfx_conv_t *h, long long *out) {
  if (!h || !h->d_prof) return fail(DFX_ERR_STATE, "no debug buffer");
  HIP_TRY(hipMemcpy(out, h->d_prof, 64 * 8, hipMemcpyDeviceToHost));
  return DFX_OK;
}
#endif

#ifdef DFX_TRACE
// diagnostic build only (make trace): host pointer to the [grid][16 waves][4] progress words
int *dfx_debug_trace(dfx_conv_t *h) { return h ? h->trace_host : nullptr; }
#endif

#ifdef DFX_STAMPS
// diagnostic build only: copies the [grid][8 waves][8] stamp sums of the last launch
int dfx_debug_read_stamps(dfx_conv_t *h, unsigned long long *out, int max_entries) {
  if (!h || !h->d_prof) return fail(DFX_ERR_STATE, "no stamps");
  int n = h->grid * (h->variant == DFX_VARIANT_MFMA_STREAM ? (h->direct ? h->nw * 16 : 96) : 768);
  if (n > max_entries) n = max_entries;
  HIP_TRY(hipMemcpy(out, h->d_prof, (size_t)n * 8, hipMemcpyDeviceToHost));
  return n;
}
#endif

int dfx_conv_destroy(dfx_conv_t *h) {
  conv_release(h);
  return DFX_OK;
}

// test hook: set (value != NULL) or clear one of the testing / tuning switches of DESIGN.md
// section 9 after the environment has been read; affects handles created afterwards
int dfx_debug_set_tuning(const char *key, const char *value) {
  if (!key) return fail(DFX_ERR_INVALID, "set_tuning: null key");
  bool known = false;
  for (const char *k : kTuningKeys) known = known || strcmp(k, key) == 0;
  if (!known) return fail(DFX_ERR_INVALID, "set_tuning: unknown switch %s", key);
  Tuning &t = tuning();
  std::lock_guard<std::mutex> lk(t.mu);
  if (value) t.kv[key] = value;
  else t.kv.erase(key);
  return DFX_OK;
}

// ---------------------------------------------------------------------------
// concat
// ---------------------------------------------------------------------------

int dfx_concat_create(const dfx_concat_desc *desc, dfx_concat_t **out) {
  if (!desc || !out || !desc->channels) return fail(DFX_ERR_INVALID, "concat_create: null argument");
  *out = nullptr;
  const dfx_concat_desc &d = *desc;
  if (d.n_inputs <= 0 || d.bs <= 0 || d.h <= 0 || d.w <= 0)
    return fail(DFX_ERR_INVALID, "concat: non-positive dimension");
  if (d.dt < DFX_F32 || d.dt > DFX_U8) return fail(DFX_ERR_INVALID, "concat: bad dtype");
  if (d.n_inputs > CONCAT_MAX_INPUTS)
    return fail(DFX_ERR_UNSUPPORTED, "concat: more than %d inputs", CONCAT_MAX_INPUTS);
  const int blk = dt_size(d.dt) == 1 ? 16 : 4;  // jit_concat_kernel.cc:155-196
  for (int i = 0; i < d.n_inputs; ++i)
    if (d.channels[i] <= 0 || d.channels[i] % blk)
      return fail(DFX_ERR_INVALID, "concat: channels of input %d not a multiple of %d", i, blk);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(DFX_ERR_NO_DEVICE, "concat_create: no HIP device (this library has no CPU path)");
  dfx_concat *h = new (std::nothrow) dfx_concat();
  if (!h) return fail(DFX_ERR_HIP, "out of host memory");
  h->d = d;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  h->channels.assign(d.channels, d.channels + d.n_inputs);
  h->d.channels = h->channels.data();
  h->d_dst = nullptr;
  h->host_stream = nullptr;
  ConcatArgs &a = h->args;
  memset(&a, 0, sizeof(a));
  const int per_chunk = 16 / (int)dt_size(d.dt);
  int end = 0;
  for (int i = 0; i < d.n_inputs; ++i) {
    end += d.channels[i] / per_chunk;
    a.chunk_end[i] = end;
  }
  a.n_inputs = d.n_inputs;
  a.chunks_per_px = end;
  a.total_chunks = (long long)d.bs * d.h * d.w * end;
  a.dt = d.dt;
  a.relu = d.post_relu;
  *out = h;
  return DFX_OK;
}

int dfx_concat_submit(dfx_concat_t *h, const void *const *srcs_dev, void *dst_dev, dfx_stream_t s) {
  if (!h || !srcs_dev || !dst_dev) return fail(DFX_ERR_INVALID, "concat_submit: null argument");
  DeviceGuard dg(h->device);
  for (int i = 0; i < h->d.n_inputs; ++i) {
    if (!srcs_dev[i]) return fail(DFX_ERR_INVALID, "concat_submit: null input %d", i);
    h->args.src[i] = (const unsigned char *)srcs_dev[i];
  }
  h->args.dst = (unsigned char *)dst_dev;
  launch_concat(h->args, (hipStream_t)s);
  HIP_TRY(hipGetLastError());
  return DFX_OK;
}

int dfx_concat_submit_gathered(dfx_concat_t *h, const void *gathered_dev, const uint64_t *offsets,
                               void *dst_dev, dfx_stream_t s) {
  if (!h || !gathered_dev || !offsets || !dst_dev)
    return fail(DFX_ERR_INVALID, "concat_submit_gathered: null argument");
  const void *ptrs[CONCAT_MAX_INPUTS];
  for (int i = 0; i < h->d.n_inputs; ++i) {
    if (offsets[i] % 16) return fail(DFX_ERR_INVALID, "concat_submit_gathered: offset %d not 16-byte aligned", i);
    ptrs[i] = (const unsigned char *)gathered_dev + offsets[i];
  }
  return dfx_concat_submit(h, ptrs, dst_dev, s);
}

int dfx_concat_submit_host(dfx_concat_t *h, const void *const *srcs_host, void *dst_host) {
  if (!h || !srcs_host || !dst_host) return fail(DFX_ERR_INVALID, "concat_submit_host: null argument");
  DeviceGuard dg(h->device);
  const size_t px = (size_t)h->d.bs * h->d.h * h->d.w, es = dt_size(h->d.dt);
  size_t oc = 0;
  for (int c : h->channels) oc += c;
  if (h->d_srcs.empty()) {
    h->d_srcs.resize(h->d.n_inputs, nullptr);
    for (int i = 0; i < h->d.n_inputs; ++i) HIP_TRY(hipMalloc(&h->d_srcs[i], px * h->channels[i] * es));
    HIP_TRY(hipMalloc(&h->d_dst, px * oc * es));
    HIP_TRY(hipStreamCreateWithFlags(&h->host_stream, hipStreamNonBlocking));
  }
  for (int i = 0; i < h->d.n_inputs; ++i)
    HIP_TRY(hipMemcpyAsync(h->d_srcs[i], srcs_host[i], px * h->channels[i] * es, hipMemcpyHostToDevice,
                           h->host_stream));
  int rc = dfx_concat_submit(h, (const void *const *)h->d_srcs.data(), h->d_dst, h->host_stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(dst_host, h->d_dst, px * oc * es, hipMemcpyDeviceToHost, h->host_stream));
  HIP_TRY(hipStreamSynchronize(h->host_stream));
  return DFX_OK;
}

int dfx_concat_destroy(dfx_concat_t *h) {
  if (!h) return DFX_OK;
  DeviceGuard dg(h->device);
  for (void *p : h->d_srcs) (void)hipFree(p);
  (void)hipFree(h->d_dst);
  if (h->host_stream) (void)hipStreamDestroy(h->host_stream);
  delete h;
  return DFX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// pooling stage of conv+relu+pool, eltwise-sum (+relu): the reference's roadmap ops
// (README.md:64-65; semantics of the MKL-DNN pipeline in test/test_conv_relu_pooling.cc:30-235)
// ---------------------------------------------------------------------------

struct dfx_pool {
  dfx_pool_desc d;
  int device;
  PoolArgs args;
};
struct dfx_eltwise {
  dfx_eltwise_desc d;
  int device;
  EltwiseArgs args;
};

int dfx_pool_create(const dfx_pool_desc *desc, dfx_pool_t **out) {
  if (!desc || !out) return fail(DFX_ERR_INVALID, "pool_create: null argument");
  *out = nullptr;
  const dfx_pool_desc &d = *desc;
  if (d.bs <= 0 || d.c <= 0 || d.ih <= 0 || d.iw <= 0 || d.oh <= 0 || d.ow <= 0 || d.kh <= 0 || d.kw <= 0 ||
      d.sh <= 0 || d.sw <= 0 || d.pad_t < 0 || d.pad_l < 0)
    return fail(DFX_ERR_INVALID, "pool: bad dimension");
  if (d.dt < DFX_F32 || d.dt > DFX_U8) return fail(DFX_ERR_INVALID, "pool: bad dtype");
  if (d.algo < DFX_POOL_MAX || d.algo > DFX_POOL_AVG_EXCLUDE_PADDING) return fail(DFX_ERR_INVALID, "pool: bad algorithm");
  // every output window must contain at least one input position (as MKL-DNN requires)
  if (d.pad_t >= d.kh || d.pad_l >= d.kw || (long long)(d.oh - 1) * d.sh - d.pad_t >= d.ih ||
      (long long)(d.ow - 1) * d.sw - d.pad_l >= d.iw)
    return fail(DFX_ERR_INVALID, "pool: an output window lies entirely in the padding");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(DFX_ERR_NO_DEVICE, "pool_create: no HIP device (this library has no CPU path)");
  dfx_pool *h = new (std::nothrow) dfx_pool();
  if (!h) return fail(DFX_ERR_HIP, "out of host memory");
  h->d = d;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  PoolArgs &a = h->args;
  memset(&a, 0, sizeof(a));
  a.bs = d.bs; a.c = d.c; a.ih = d.ih; a.iw = d.iw; a.oh = d.oh; a.ow = d.ow;
  a.kh = d.kh; a.kw = d.kw; a.sh = d.sh; a.sw = d.sw; a.pad_t = d.pad_t; a.pad_l = d.pad_l; a.dt = d.dt;
  a.algo = d.algo;
  const size_t es = dt_size(d.dt);
  a.vec = ((size_t)d.c * es) % 16 == 0;
  a.groups = a.vec ? (int)((size_t)d.c * es / 16) : d.c;
  a.total = (long long)d.bs * d.oh * d.ow * a.groups;
  *out = h;
  return DFX_OK;
}

int dfx_pool_submit(dfx_pool_t *h, const void *src_dev, void *dst_dev, dfx_stream_t s) {
  if (!h || !src_dev || !dst_dev) return fail(DFX_ERR_INVALID, "pool_submit: null argument");
  DeviceGuard dg(h->device);
  PoolArgs a = h->args;  // per-launch copy: concurrent submits on several streams are independent
  a.src = (const unsigned char *)src_dev;
  a.dst = (unsigned char *)dst_dev;
  if (a.vec && (((uintptr_t)src_dev | (uintptr_t)dst_dev) % 16)) {  // 16-byte vector path needs aligned buffers:
    a.vec = 0;                                                       // misaligned ones take the per-channel path
    a.groups = a.c;
    a.total = (long long)a.bs * a.oh * a.ow * a.groups;
  }
  if (launch_pool(a, (hipStream_t)s) != 0) return fail(DFX_ERR_INVALID, "pool_submit: bad dtype");
  HIP_TRY(hipGetLastError());
  return DFX_OK;
}

int dfx_pool_destroy(dfx_pool_t *h) {
  delete h;
  return DFX_OK;
}

int dfx_eltwise_create(const dfx_eltwise_desc *desc, dfx_eltwise_t **out) {
  if (!desc || !out) return fail(DFX_ERR_INVALID, "eltwise_create: null argument");
  *out = nullptr;
  const dfx_eltwise_desc &d = *desc;
  if (d.n_inputs < 2 || d.n_inputs > ELTWISE_MAX_INPUTS)
    return fail(DFX_ERR_UNSUPPORTED, "eltwise: 2..%d inputs", ELTWISE_MAX_INPUTS);
  if (d.elems <= 0) return fail(DFX_ERR_INVALID, "eltwise: non-positive size");
  if (d.dt < DFX_F32 || d.dt > DFX_U8) return fail(DFX_ERR_INVALID, "eltwise: bad dtype");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(DFX_ERR_NO_DEVICE, "eltwise_create: no HIP device (this library has no CPU path)");
  dfx_eltwise *h = new (std::nothrow) dfx_eltwise();
  if (!h) return fail(DFX_ERR_HIP, "out of host memory");
  h->d = d;
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  memset(&h->args, 0, sizeof(h->args));
  h->args.n_inputs = d.n_inputs;
  h->args.dt = d.dt;
  h->args.relu = d.post_relu;
  h->args.elems = d.elems;
  *out = h;
  return DFX_OK;
}

int dfx_eltwise_submit(dfx_eltwise_t *h, const void *const *srcs_dev, void *dst_dev, dfx_stream_t s) {
  if (!h || !srcs_dev || !dst_dev) return fail(DFX_ERR_INVALID, "eltwise_submit: null argument");
  DeviceGuard dg(h->device);
  EltwiseArgs a = h->args;
  for (int i = 0; i < a.n_inputs; ++i) {
    if (!srcs_dev[i]) return fail(DFX_ERR_INVALID, "eltwise_submit: null input %d", i);
    if ((uintptr_t)srcs_dev[i] % 16) return fail(DFX_ERR_INVALID, "eltwise_submit: input %d not 16-byte aligned", i);
    a.src[i] = (const unsigned char *)srcs_dev[i];
  }
  if ((uintptr_t)dst_dev % 16) return fail(DFX_ERR_INVALID, "eltwise_submit: dst not 16-byte aligned");
  a.dst = (unsigned char *)dst_dev;
  if (launch_eltwise(a, (hipStream_t)s) != 0) return fail(DFX_ERR_INVALID, "eltwise_submit: bad dtype");
  HIP_TRY(hipGetLastError());
  return DFX_OK;
}

int dfx_eltwise_destroy(dfx_eltwise_t *h) {
  delete h;
  return DFX_OK;
}
