/*
 * dfx.h -- C ABI of the MI355X (gfx950) implementation of deep-fusion's hot path:
 * the fused int8 conv3x3+relu+conv1x1(+relu) primitive and the concat(+relu)
 * primitive.  Plain pointers and sizes only; no C++/torch types cross this
 * boundary.  The library behind it is libdfx_hip.so (deep-fusion_amd/csrc/).
 *
 * This boundary sits where the reference has
 *     jit_conv_conf_t / jit_conv_call_t / void (*jit_ker_)(jit_conv_call_t*)
 *     (/root/reference/src/jit_call_conf.h:48-99, src/jit_conv_kernel.h:50-51)
 * and the same trio for concat (jit_call_conf.h:29-45, jit_concat_kernel.h:38-39):
 * a create-time POD descriptor, per-call buffer pointers, and one entry point
 * that runs the kernel.  Differences, by design:
 *   - one call per submit (the reference calls the JIT kernel once per output
 *     row, op_conv.cc:217-238); the call enqueues on a HIP stream;
 *   - weights / bias / scales are copied at dfx_conv_set_weights() and owned by
 *     the handle (fixes the dangling scales pointer, op_conv.h:94-95);
 *   - every entry point returns an int status (0 = ok) instead of exit()ing
 *     (log.h:38-42); dfx_last_error() returns the message.
 *
 * There is no CPU fallback: every compute entry point fails with
 * DFX_ERR_NO_DEVICE / DFX_ERR_HIP when no gfx950 device is usable.
 */
#ifndef DFX_H
#define DFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_VERSION 100

/* status codes */
enum {
  DFX_OK = 0,
  DFX_ERR_INVALID = 1,     /* descriptor violates the reference's shape/dtype rules */
  DFX_ERR_UNSUPPORTED = 2, /* valid for the reference, not implemented here */
  DFX_ERR_HIP = 3,         /* a HIP runtime call failed */
  DFX_ERR_NO_DEVICE = 4,   /* no usable gfx950 device */
  DFX_ERR_STATE = 5        /* e.g. submit before set_weights */
};

/* values of deepfusion::memory::dtype (reference include/deepfusion.h:66-72) */
enum { DFX_UNDEF = 0, DFX_F32 = 1, DFX_S32 = 2, DFX_S8 = 3, DFX_U8 = 4 };
/* deepfusion::round_mode (include/deepfusion.h:46-49) */
enum { DFX_ROUND_NEAREST = 0, DFX_ROUND_DOWN = 1 };

/* kernel variants (dfx_conv_info.variant) */
enum {
  DFX_VARIANT_GENERIC = 0,    /* any shape the reference's init_conf accepts */
  DFX_VARIANT_MFMA_FUSED = 1, /* int8-MFMA implicit GEMM, 3x3 s1 + fused 1x1 */
  DFX_VARIANT_MFMA_CONV = 2,  /* int8-MFMA implicit GEMM, unfused 3x3 s1 conv */
  DFX_VARIANT_MFMA_STREAM = 3 /* int8-MFMA, streamed weights: any kernel/stride/channel count */
};

/* Create-time descriptor.  Mirrors the shape/dtype/flag fields of
 * jit_conv_conf_t (jit_call_conf.h:66-99); the x86 blocking fields
 * (nb_*_blocking, ur_w, typesize_*, use_vnni) are dropped.
 * Tensors: src NHWC u8 {bs,ih,iw,ic}; wei OIhw4i16o4i s8 {oc,ic,kh,kw};
 * wei1x1 OIhw4i16o4i s8 {oc1x1,oc,1,1}; bias format x; dst NHWC. */
typedef struct dfx_conv_desc {
  int32_t bs;
  int32_t ic, ih, iw;
  int32_t oc, oh, ow;      /* conv0 output; oh/ow must equal (in+2p-k)/s+1 */
  int32_t kh, kw, sh, sw;
  int32_t pad_t, pad_l;
  int32_t oc1x1;           /* 0 = unfused conv (deepfusion.h:121-129) */
  int32_t dst_dt;          /* DFX_F32 | DFX_S32 | DFX_S8 | DFX_U8 */
  int32_t bia0_dt;         /* DFX_UNDEF = no bias */
  int32_t bia1_dt;
  int32_t conv0_relu, conv1_relu;
  int32_t conv0_round_mode, conv1_round_mode;
  int32_t conv0_nscales;   /* 1 or oc     (op_conv.cc:311-313) */
  int32_t conv1_nscales;   /* 1 or oc1x1  (op_conv.cc:342-345) */
  int32_t force_variant;   /* -1 = auto, else DFX_VARIANT_* (testing) */
  int32_t fuse_pool;       /* 0 = none; 2 = 2x2 stride-2 max pooling of the conv's output (after ReLU and
                              requantisation) fused into the conv kernel: dst then holds {bs, oh/2, ow/2, oc}.
                              Only for unfused 3x3 stride-1 convs on the resident-weight kernel with even oh, ow;
                              dfx_conv_create returns DFX_ERR_UNSUPPORTED otherwise (the caller then runs
                              dfx_pool_* behind an unpooled conv, as deepfusion::conv_relu_pool does). */
} dfx_conv_desc;

typedef struct dfx_conv_info {
  int32_t variant;
  int32_t grid, block, lds_bytes;
  int32_t rows_per_unit;       /* output rows one workgroup produces */
  int32_t device;              /* ordinal of the device the handle lives on */
  uint64_t algorithmic_ops;    /* 2*MAC of one submit */
  uint64_t algorithmic_bytes;  /* src + weights + dst bytes of one submit */
  char kernel_name[96];
} dfx_conv_info;

typedef struct dfx_concat_desc {
  int32_t n_inputs;
  int32_t bs, h, w;
  int32_t dt;               /* all inputs and dst share it (jit_concat_kernel.cc:184-187) */
  int32_t post_relu;
  const int32_t *channels;  /* n_inputs entries; %16 (1-byte) or %4 (4-byte) */
} dfx_concat_desc;

typedef struct dfx_conv dfx_conv_t;
typedef struct dfx_concat dfx_concat_t;

/* ---- the reference's roadmap ops (README.md:64-65: "conv+relu+pooling fused op",
 *      "eltwise-sum + relu fused op"; planned signature in test/test_conv_relu_pooling.cc:264-281).
 *      The reference ships no implementation: semantics follow the MKL-DNN pipeline its test
 *      builds (test_conv_relu_pooling.cc:30-235).  Parity unpinned. ---- */
typedef enum dfx_pool_algo {
  DFX_POOL_MAX = 0,
  DFX_POOL_AVG_INCLUDE_PADDING = 1,  /* sum over the window's positions inside the input / (kh * kw) */
  DFX_POOL_AVG_EXCLUDE_PADDING = 2   /* ... / number of those positions */
} dfx_pool_algo;
typedef struct dfx_pool_desc {
  int32_t bs, c, ih, iw;    /* NHWC input (the conv's output) */
  int32_t oh, ow;           /* given by the caller like the reference's dst_dims; windows may hang over the
                               bottom / right edge (the test's padR search, test_conv_relu_pooling.cc:178-182) */
  int32_t kh, kw, sh, sw, pad_t, pad_l;
  int32_t dt;               /* dfx_dtype of src and dst */
  int32_t algo;             /* DFX_POOL_MAX: maximum over the window's positions INSIDE the input
                               (padding does not take part, as in MKL-DNN's pooling_max).
                               DFX_POOL_AVG_*: integer types sum exactly, divide in f32 (float(sum) /
                               float(count)), round to nearest even and saturate to the dtype; f32 sums in
                               window order and divides (the reference test's pooling_avg_* flags,
                               test_conv_relu_pooling.cc:189-193; MKL-DNN's reference pooling arithmetic) */
} dfx_pool_desc;
typedef struct dfx_pool dfx_pool_t;

typedef struct dfx_eltwise_desc {
  int32_t n_inputs;         /* 2..8 tensors of identical shape and dtype */
  int64_t elems;            /* elements per tensor */
  int32_t dt;
  int32_t post_relu;        /* dst = relu?(saturate(sum_i src_i)): integer sums are exact and saturate to the
                               dtype's range, f32 sums left to right */
} dfx_eltwise_desc;
typedef struct dfx_eltwise dfx_eltwise_t;
typedef void *dfx_stream_t; /* a hipStream_t; NULL = the default stream */
typedef void *dfx_event_t;  /* a hipEvent_t */

/* ---- library / device ---- */
int dfx_version(void);
const char *dfx_last_error(void);        /* thread-local message of the last failure */
int dfx_device_count(int *count);
int dfx_set_device(int ordinal);
int dfx_device_name(char *buf, size_t len);

/* ---- buffers and streams: replace util/memory.cc:21-40 (aligned_malloc/free)
 *      and util/omp_thread.h:18-25 (the OpenMP shim) ---- */
int dfx_mem_alloc_host(void **p, size_t bytes);    /* pinned host memory */
int dfx_mem_free_host(void *p);
int dfx_mem_alloc_device(void **p, size_t bytes);
int dfx_mem_free_device(void *p);
int dfx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, dfx_stream_t s);
int dfx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, dfx_stream_t s);
int dfx_memset_device(void *dst_dev, int value, size_t bytes, dfx_stream_t s);
int dfx_stream_create(dfx_stream_t *s);
int dfx_stream_destroy(dfx_stream_t s);
int dfx_stream_sync(dfx_stream_t s);
/* work enqueued on `waiter` after this call starts only when everything enqueued on `producer`
 * before it has finished (chains of asynchronous submits on different streams) */
int dfx_stream_wait_stream(dfx_stream_t waiter, dfx_stream_t producer);
/* device-side timing (the DEEPFUSION_PROFILE hook of op::submit, deepfusion.cc:91-102) */
int dfx_event_create(dfx_event_t *e);
int dfx_event_record(dfx_event_t e, dfx_stream_t s);
int dfx_event_elapsed_ms(dfx_event_t start, dfx_event_t stop, float *ms); /* waits for `stop` */
int dfx_event_destroy(dfx_event_t e);

/* ---- weight reorder (the reference exposes OIhw4i16o4i, deepfusion.h:59-60,
 *      but ships no reorder, deepfusion.cc:44-50).  Host-side, pure layout. ---- */
int dfx_reorder_oihw_to_blocked(const int8_t *oihw, int8_t *blocked, int O, int I,
                                int KH, int KW);
size_t dfx_blocked_offset(int o, int i, int kh, int kw, int I, int KH, int KW);

/* ---- conv: replaces op_conv<T> (src/op_conv.h:34-96, src/op_conv.cc:31-260)
 *      and jit_conv_kernel (src/jit_conv_kernel.cc:27-510) ---- */
/* validates like op_conv<T>::init_conf + jit_conv_kernel::init_conf
 * (op_conv.cc:262-365, jit_conv_kernel.cc:512-673) minus the defects of
 * SURVEY.md 8(a); picks a kernel variant. */
int dfx_conv_create(const dfx_conv_desc *desc, dfx_conv_t **out);
/* A handle lives on the device that was current (dfx_set_device) when it was created; every
 * later entry point switches to that device for the duration of the call, so one host thread can
 * drive handles on several devices. */
/* host pointers; data is copied (and repacked for the MFMA variant) into
 * device memory owned by the handle.  wei1x1/bia1x1/scales1 are ignored for an
 * unfused op; bias pointers may be NULL when the dtype is DFX_UNDEF. */
int dfx_conv_set_weights(dfx_conv_t *h, const int8_t *wei_blocked, const void *bia0,
                         const float *scales0, const int8_t *wei1x1_blocked,
                         const void *bia1, const float *scales1);
/* asynchronous: enqueues on `s`; src_dev/dst_dev are device pointers that must
 * stay valid until the stream reaches the kernel's end.  A handle may be submitted from several
 * host threads and on several streams at once: every launch works on its own copy of the
 * arguments and its own unit-queue slot.  There are 16 slots per handle: launches on one stream are
 * ordered anyway; with several streams a launch that finds its slot last used on ANOTHER stream first
 * waits, on the device, for that launch (a 17th concurrent launch queues behind the 1st).  Every op is ONE
 * kernel launch (the two-launch "split:" ops of earlier versions are gone); dfx_conv_info.kernel_name names
 * the kernel, for the role-specialised one also its stage-1 requant route ("/fma", "/magic"), and says so
 * when an op runs on the scalar kernel because its dst reaches 4 GiB. */
int dfx_conv_submit(dfx_conv_t *h, const void *src_dev, void *dst_dev, dfx_stream_t s);
/* drop-in semantics of op::submit() (deepfusion.cc:90-103): host buffers in,
 * host buffers out, synchronous (H2D, kernel, D2H, stream sync). */
int dfx_conv_submit_host(dfx_conv_t *h, const void *src_host, void *dst_host);
int dfx_conv_query(const dfx_conv_t *h, dfx_conv_info *info);
int dfx_conv_destroy(dfx_conv_t *h);

/* ---- concat: replaces op_concat<T> (src/op_concat.h:28-61, op_concat.cc:22-72)
 *      and jit_concat_kernel (src/jit_concat_kernel.cc:30-197) ---- */
int dfx_concat_create(const dfx_concat_desc *desc, dfx_concat_t **out);
int dfx_concat_submit(dfx_concat_t *h, const void *const *srcs_dev, void *dst_dev,
                      dfx_stream_t s);
int dfx_concat_submit_host(dfx_concat_t *h, const void *const *srcs_host, void *dst_host);
/* Concat of channel slices that live in one rank-major staging buffer, as left
 * by an all-gather of per-rank NHWC shards {bs,h,w,channels[r]}:
 * input r starts at byte offset offsets[r] of `gathered_dev`. */
int dfx_concat_submit_gathered(dfx_concat_t *h, const void *gathered_dev,
                               const uint64_t *offsets, void *dst_dev, dfx_stream_t s);
int dfx_concat_destroy(dfx_concat_t *h);

/* ---- pooling stage of conv+relu+pool, eltwise-sum(+relu) (see the descriptors above) ---- */
int dfx_pool_create(const dfx_pool_desc *desc, dfx_pool_t **out);
int dfx_pool_submit(dfx_pool_t *h, const void *src_dev, void *dst_dev, dfx_stream_t s);
int dfx_pool_destroy(dfx_pool_t *h);
int dfx_eltwise_create(const dfx_eltwise_desc *desc, dfx_eltwise_t **out);
int dfx_eltwise_submit(dfx_eltwise_t *h, const void *const *srcs_dev, void *dst_dev, dfx_stream_t s);
int dfx_eltwise_destroy(dfx_eltwise_t *h);

/* ---- test hooks (not part of the reference's surface; used by tests/ only) ---- */
/* Overwrites the LDS of every CU with a pattern (asynchronous, on `s`): makes a kernel that
 * reads LDS before publishing it fail deterministically (tests/test_gpu_first_launch.py). */
int dfx_debug_scribble_lds(unsigned pattern, dfx_stream_t s);
/* Sets (value != NULL) or clears a testing / tuning switch (DFX_NO_FAST,
 * DFX_FORCE_GEOM, DFX_STREAM_GRID ...).  The library reads those from the environment once, when
 * it is first used; this is how a test reaches another code path afterwards.  Affects handles
 * created after the call. */
int dfx_debug_set_tuning(const char *key, const char *value);

#ifdef __cplusplus
}
#endif
#endif
