// deepfusion.h -- drop-in C++ API of the MI355X (gfx950) build of deep-fusion.
//
// Source-compatible with the public header of the reference
// (/root/reference/include/deepfusion.h): the same namespace, element typedefs,
// `round_mode`, `memory` (formats, dtypes, both constructors, accessors) and the
// same `op::submit()`, `concat()` and two `conv()` entry points with identical
// parameter lists and defaults, so callers such as the reference's
// test/test_concat.cc:89-108 or benchmark/bench_concat.cc:124-161 compile
// unchanged.  Behind it the Xbyak JIT kernels are replaced by HIP kernels reached
// through the C ABI of include/dfx.h.
//
// Host/device coherence (the reference has one address space; this build has two):
//   * memory::data() still returns a HOST pointer that callers read and write
//     directly.  The buffer is pinned host memory.
//   * submit() has the reference's semantics: the caller's host buffers are re-read on EVERY
//     call.  Inputs are uploaded every time (also when the caller refilled them through a
//     pointer fetched once, without calling data() again), borrowed weight / bias tensors are
//     hashed and re-packed when their bytes changed; then ONE kernel is launched, the
//     destination downloaded and the stream synchronised: after submit() returns,
//     dst->data() holds the result, exactly like the reference's synchronous OpenMP
//     execution (deepfusion.cc:90-103).  The one exception: an input whose device copy was
//     written by another op's submit_async() and that nobody has touched through data()
//     since is NOT uploaded -- its host bytes are the stale side.
//   * Extensions for device-resident pipelines (not in the reference):
//     op::submit_async() skips the download and the synchronisation and trusts the
//     per-tensor data() counters instead of re-reading host memory (every call of data()
//     marks the tensor "host-dirty": it is uploaded / its weights are re-packed by the next
//     submit_async()); memory::device_data() exposes the device buffer; memory::download() /
//     op::wait() complete a transfer explicitly.  With DEEPFUSION_PROFILE=1 every submit,
//     submit_async() included, waits for its launch (it prints the launch's duration).
//
// Lifetime rule kept from the reference (op_conv.h:81-95, op_concat.h:53-56):
// an op borrows the tensors it was built from; they must outlive it.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <array>
#include <memory>
#include <vector>

namespace deepfusion {

typedef float f32;
typedef int32_t s32;
typedef int8_t s8;
typedef uint8_t u8;

// kept for source compatibility with code that uses the reference's macro
#ifndef DISABLE_COPY_AND_ASSIGN
#define DISABLE_COPY_AND_ASSIGN(classname)          \
private:                                            \
  classname(const classname &) = delete;            \
  classname(const classname &&) = delete;           \
  classname &operator=(const classname &) = delete; \
  classname &operator=(const classname &&) = delete
#endif

struct opdesc {  // placeholder type of the reference API (deepfusion.h:42-44), unused
  int tmp;
};

// rounding of the f32 -> integer conversion of a requantisation stage
enum round_mode {
  nearest = 0,  // ties to even (vcvtps2dq rn-sae)
  down,         // toward -inf  (vcvtps2dq rd-sae)
};

namespace detail {
struct memory_state;  // pinned host buffer, lazily created device buffer, dirty flags
struct op_state;
}  // namespace detail

struct memory {
public:
  // physical layouts.  nhwc: activations.  OIhw4i16o4i: s8 weights blocked as
  // [O/16][I/16][kh][kw][(i%16)/4][o%16][i%4] (see dfx_reorder_oihw_to_blocked in
  // dfx.h and reorder_weights() below).  x: 1-D (bias).
  enum format {
    format_undef = 0,
    x,
    nchw,
    oihw = nchw,
    nhwc,
    OIhw4i16o4i,
    gOIhw4i16o4i,
  };
  typedef std::vector<int> dims;
  typedef std::array<int, 2> pair_dims;
  typedef std::array<int, 4> nchw_dims;

  enum dtype {
    undef = 0,
    f32,
    s32,
    s8,
    u8,
  };

  // logical nchw / oihw dims; the physical order follows `fmt`
  explicit memory(const nchw_dims &dm, const format fmt, const dtype dt, int alignment = 4096);
  // physical dims as given (used for format x)
  explicit memory(const dims &dm, const format fmt, const dtype dt, int alignment = 4096);
  ~memory();

  size_t size();         // number of elements
  size_t buffer_size();  // bytes
  dims actual_dims() { return dims_; }
  nchw_dims std_dims() { return std_dims_; }  // nchw or oihw
  dtype data_type() { return dt_; }
  format dim_format() { return fmt_; }
  void *data();  // host pointer; marks the tensor host-dirty

  // ---- extensions of this build ----
  const void *host_data() const;  // host pointer without marking it dirty
  void *device_data();            // device buffer (allocated on first use)
  void upload();                  // host -> device now (clears host-dirty)
  void download();                // device -> host now, synchronous
  unsigned long long host_version() const;  // incremented by every data() call

private:
  friend struct detail::op_state;
  void allocate_buffer(int alignment);
  detail::memory_state *st_;
  dims dims_;
  nchw_dims std_dims_;
  format fmt_;
  dtype dt_;

  DISABLE_COPY_AND_ASSIGN(memory);
};

class op {
public:
  explicit op() {}
  virtual ~op() {}
  // synchronous, like the reference: upload dirty inputs, run, download dst, wait
  virtual void submit();
  // extension: enqueue only (inputs uploaded if dirty); result stays on the device
  virtual void submit_async();
  // extension: block until everything this op enqueued has finished
  virtual void wait();

protected:
  virtual void infer() = 0;
  virtual const char *name() = 0;
  DISABLE_COPY_AND_ASSIGN(op);
};

// channel concat of nhwc tensors with optional ReLU (reference deepfusion.h:116-118)
std::unique_ptr<op> concat(const std::vector<std::unique_ptr<memory>> &srcs,
                           std::unique_ptr<memory> &dst,
                           bool post_relu = false);

// convolution only (reference deepfusion.h:121-129)
std::unique_ptr<op> conv(const std::unique_ptr<memory> &src,
                         const std::unique_ptr<memory> &wei,
                         const std::unique_ptr<memory> &bia,
                         std::array<int, 2> sz_stride,
                         std::array<int, 2> sz_padding,
                         std::unique_ptr<memory> &dst,
                         bool conv0_relu = false,
                         std::vector<float> conv0_scales = {1.f},
                         round_mode conv0_round_mode = round_mode::nearest);

// convolution fused with relu + 1x1 convolution (+relu) (reference deepfusion.h:132-145)
std::unique_ptr<op> conv(const std::unique_ptr<memory> &src,
                         const std::unique_ptr<memory> &wei,
                         const std::unique_ptr<memory> &bia,
                         std::array<int, 2> sz_stride,
                         std::array<int, 2> sz_padding,
                         const std::unique_ptr<memory> &wei1x1,
                         const std::unique_ptr<memory> &bia1x1,
                         std::unique_ptr<memory> &dst,
                         bool conv0_relu = false,
                         std::vector<float> conv0_scales = {1.f},
                         round_mode conv0_round_mode = round_mode::nearest,
                         bool conv1_relu = false,
                         std::vector<float> conv1_scales = {1.f},
                         round_mode conv1_round_mode = round_mode::nearest);

// ---- the reference's roadmap ops (README.md:64-65), which it never shipped.  Signature after the
// planned one in test/test_conv_relu_pooling.cc:264-281; semantics of the MKL-DNN pipeline that test
// builds (:30-235): conv (+bias, scale, round) -> relu -> max pooling whose padding takes no part.
// `dst` carries the pooled dims; the conv output dims follow from src / wei / stride / padding. ----
enum class pool_algo {  // the reference test's max_pooling / pooling_avg_* flags (test_conv_relu_pooling.cc:189-193)
  max = 0,
  avg_include_padding,
  avg_exclude_padding,
};
std::unique_ptr<op> conv_relu_pool(const std::unique_ptr<memory> &src,
                                   const std::unique_ptr<memory> &wei,
                                   const std::unique_ptr<memory> &bia,
                                   std::array<int, 2> conv_stride,
                                   std::array<int, 2> conv_padding,
                                   std::array<int, 2> pool_kernel,
                                   std::array<int, 2> pool_stride,
                                   std::array<int, 2> pool_padding,
                                   std::unique_ptr<memory> &dst,
                                   bool conv_relu = true,
                                   std::vector<float> conv_scales = {1.f},
                                   round_mode conv_round_mode = round_mode::nearest,
                                   pool_algo algo = pool_algo::max);

// dst = relu?(saturate(sum of srcs)): same shape, format and dtype everywhere; integer sums are
// exact and saturate to the dtype's range, f32 sums run left to right (README.md:65, the "shortcut
// sum" of test_conv_relu_pooling.cc:118-124)
std::unique_ptr<op> eltwise_sum(const std::vector<std::unique_ptr<memory>> &srcs,
                                std::unique_ptr<memory> &dst,
                                bool post_relu = true);

// ---- extension: the reorder the reference never shipped (deepfusion.cc:44-50) ----
// Writes plain oihw s8 weights into `blocked` (an OIhw4i16o4i memory of the same
// logical dims) in the [O/16][I/16][kh][kw][4i][16o][4i] byte order.
void reorder_weights(const s8 *oihw, const std::unique_ptr<memory> &blocked);

}  // namespace deepfusion
