// bench_conv -- times the fused conv3x3+relu+conv1x1(+relu) (or the unfused conv when
// -oc1x1 0) through the drop-in C++ API.  Flag names follow the reference's stub
// benchmark/bench_conv.cc:22-37 (which parses them and returns); the timing protocol
// follows benchmark/bench_concat.cc:141-159.
//   bench_conv -bs 128 -ih 56 -iw 56 -kh 3 -kw 3 -sh 1 -sw 1 -ph 1 -pw 1 -ic 64 -oc 64 -oc1x1 256 -dtype s32
#include <chrono>
#include <cstdio>

#include "cli_flags.h"
#include "deepfusion.h"
#include "dfx.h"

using namespace deepfusion;

// -cold_cache (the reference's WITH_COLD_CACHE build option, CMakeLists.txt:60-61, test/test_utils.cc:23-45: a
// scratch buffer is rewritten around each timed call and the per-call times are averaged, bench_concat.cc:141-159).
// Here: 512 MiB of device scratch (twice the 256 MiB Infinity Cache) is rewritten before EVERY timed launch, each
// launch is timed on its own (host clock around submit_async() + wait()) and the mean is reported, next to the
// same per-launch protocol without the flush so that the two are comparable.
template <typename Op>
static void cold_cache_leg(Op &op, int iters, const char *what) {
  const size_t scratch_bytes = 512u << 20;
  void *scratch = nullptr;
  if (dfx_mem_alloc_device(&scratch, scratch_bytes) != DFX_OK) { fprintf(stderr, "cold_cache: %s\n", dfx_last_error()); exit(1); }
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double sum[2] = {0, 0};
  for (int cold = 0; cold < 2; ++cold)
    for (int i = 0; i < iters; ++i) {
      if (cold) {
        dfx_memset_device(scratch, i & 0xff, scratch_bytes, nullptr);
        dfx_stream_sync(nullptr);
      }
      const double t0 = now();
      op->submit_async();
      op->wait();
      sum[cold] += now() - t0;
    }
  dfx_mem_free_device(scratch);
  printf("DeepFusion %s avg time (device resident, one launch at a time, warm caches): %f ms\n", what, sum[0] / iters);
  printf("DeepFusion %s avg time (device resident, one launch at a time, COLD caches: 512 MiB scratch rewritten before each): %f ms\n", what, sum[1] / iters);
}

int main(int argc, char **argv) {
  Flags f(argc, argv);
  const int burn = f.geti("burning_iter", 50), iters = f.geti("iter", 100);
  const int bs = f.geti("bs", 128), ih = f.geti("ih", 56), iw = f.geti("iw", 56);
  const int kh = f.geti("kh", 3), kw = f.geti("kw", 3), sh = f.geti("sh", 1), sw = f.geti("sw", 1);
  const int ph = f.geti("ph", 1), pw = f.geti("pw", 1);
  const int ic = f.geti("ic", 64), oc = f.geti("oc", 64), oc1 = f.geti("oc1x1", 256);
  const std::string dts = f.gets("dtype", "s32");
  const bool relu = f.getb("post_relu", true);
  memory::dtype dt = dts == "f32" ? memory::dtype::f32 : dts == "s32" ? memory::dtype::s32
                   : dts == "s8" ? memory::dtype::s8 : memory::dtype::u8;
  const int oh = (ih + 2 * ph - kh) / sh + 1, ow = (iw + 2 * pw - kw) / sw + 1;
  Lcg g(1234);
  std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
  std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, kh, kw}, memory::format::OIhw4i16o4i, memory::dtype::s8));
  std::unique_ptr<memory> bia(new memory(memory::dims{oc}, memory::format::x, memory::dtype::s32));
  std::unique_ptr<memory> wei1, bia1;
  { uint8_t *p = (uint8_t *)src->data(); for (size_t i = 0; i < src->size(); ++i) p[i] = (uint8_t)(g.next() % 17); }
  { std::vector<s8> w(wei->size()); for (auto &v : w) v = (s8)((int)(g.next() % 21) - 10); reorder_weights(w.data(), wei); }
  { int32_t *p = (int32_t *)bia->data(); for (int i = 0; i < oc; ++i) p[i] = (int)(g.next() % 21) - 10; }
  if (oc1 > 0) {
    wei1.reset(new memory(memory::nchw_dims{oc1, oc, 1, 1}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    bia1.reset(new memory(memory::dims{oc1}, memory::format::x, memory::dtype::s32));
    std::vector<s8> w(wei1->size()); for (auto &v : w) v = (s8)((int)(g.next() % 21) - 10); reorder_weights(w.data(), wei1);
    int32_t *p = (int32_t *)bia1->data(); for (int i = 0; i < oc1; ++i) p[i] = (int)(g.next() % 21) - 10;
  }
  std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc1 > 0 ? oc1 : oc, oh, ow}, memory::format::nhwc, dt));
  std::unique_ptr<op> c = oc1 > 0
      ? conv(src, wei, bia, {sh, sw}, {ph, pw}, wei1, bia1, dst, true, {1.f / 64}, round_mode::nearest, relu, {1.f / 16}, round_mode::nearest)
      : conv(src, wei, bia, {sh, sw}, {ph, pw}, dst, relu, {1.f / 64}, round_mode::nearest);
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int i = 0; i < burn; ++i) c->submit_async();
  c->wait();
  double t0 = now();
  for (int i = 0; i < iters; ++i) c->submit_async();
  c->wait();
  const double dev_ms = (now() - t0) / iters;
  c->submit();  // warm
  t0 = now();
  const int hi = iters > 10 ? 10 : iters;
  for (int i = 0; i < hi; ++i) c->submit();
  const double host_ms = (now() - t0) / hi;
  printf("Conv bs=%d %dx%d ic=%d oc=%d oc1x1=%d k=%dx%d s=%d p=%d dst=%s\n", bs, ih, iw, ic, oc, oc1, kh, kw, sh, ph, dts.c_str());
  printf("DeepFusion Conv avg time (device resident):            %f ms  (%.1f images/sec)\n", dev_ms, bs / dev_ms * 1e3);
  printf("DeepFusion Conv avg time (submit: H2D + kernel + D2H): %f ms  (%.1f images/sec)\n", host_ms, bs / host_ms * 1e3);
  if (f.getb("cold_cache", false)) cold_cache_leg(c, iters, "Conv");
  return 0;
}
