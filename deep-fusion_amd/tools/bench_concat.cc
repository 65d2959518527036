// bench_concat -- times deepfusion::concat on MI355X through the drop-in C++ API.
// Flag names and the timing protocol follow the reference's
// benchmark/bench_concat.cc:22-29, :124-161 (burning_iter warm-up submits, iter timed
// submits, mean ms); the MKL-DNN comparison leg is gone (dependency unavailable).
//   bench_concat -n 4 -c 64,96 -h 64 -w 64 -dtype s8 -post_relu
#include <chrono>
#include <cmath>
#include <cstdio>

#include "cli_flags.h"
#include "deepfusion.h"
#include "dfx.h"

using namespace deepfusion;

static memory::dtype parse_dt(const std::string &s) {
  if (s == "f32") return memory::dtype::f32;
  if (s == "s32") return memory::dtype::s32;
  if (s == "s8") return memory::dtype::s8;
  if (s == "u8") return memory::dtype::u8;
  fprintf(stderr, "Unknow data type %s\n", s.c_str());
  exit(1);
}

template <typename T>
static void fill(void *p, size_t n, memory::dtype dt, Lcg &g) {
  T *d = static_cast<T *>(p);
  for (size_t i = 0; i < n; ++i) {
    if (dt == memory::dtype::f32) d[i] = (T)(1.0 + 0.01 * sinf((float)(i % 37)));
    else if (dt == memory::dtype::u8) d[i] = (T)(g.next() % 17);
    else d[i] = (T)((int)(g.next() % 21) - 10);
  }
}

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
  const int n = f.geti("n", 4), h = f.geti("h", 64), w = f.geti("w", 64);
  std::vector<int> ch = Flags::split_ints(f.gets("c", "64,96"));
  const memory::dtype dt = parse_dt(f.gets("dtype", "s8"));
  const bool relu = f.getb("post_relu", true);
  Lcg g(1234);
  std::vector<std::unique_ptr<memory>> srcs;
  int oc = 0;
  for (int c : ch) {
    srcs.emplace_back(new memory(memory::nchw_dims{n, c, h, w}, memory::format::nhwc, dt));
    memory &m = *srcs.back();
    if (dt == memory::dtype::f32) fill<float>(m.data(), m.size(), dt, g);
    else if (dt == memory::dtype::s32) fill<int32_t>(m.data(), m.size(), dt, g);
    else if (dt == memory::dtype::s8) fill<int8_t>(m.data(), m.size(), dt, g);
    else fill<uint8_t>(m.data(), m.size(), dt, g);
    oc += c;
  }
  std::unique_ptr<memory> dst(new memory(memory::nchw_dims{n, oc, h, w}, memory::format::nhwc, dt));
  auto op = concat(srcs, dst, relu);
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int i = 0; i < burn; ++i) op->submit();
  double t0 = now();
  for (int i = 0; i < iters; ++i) op->submit();
  double host_ms = (now() - t0) / iters;
  for (int i = 0; i < burn; ++i) op->submit_async();
  op->wait();
  t0 = now();
  for (int i = 0; i < iters; ++i) op->submit_async();
  op->wait();
  double dev_ms = (now() - t0) / iters;
  const double bytes = 2.0 * dst->buffer_size();
  printf("Concat %d inputs -> {%d,%d,%d,%d} %s relu=%d\n", (int)ch.size(), n, oc, h, w, f.gets("dtype", "s8").c_str(), relu);
  printf("DeepFusion Concat avg time (submit: H2D + kernel + D2H): %f ms\n", host_ms);
  printf("DeepFusion Concat avg time (device resident):            %f ms  (%.1f GB/s)\n", dev_ms, bytes / dev_ms / 1e6);
  if (f.getb("cold_cache", false)) cold_cache_leg(op, iters, "Concat");
  return 0;
}
