// dropin_check -- exercises the drop-in C++ API (include/deepfusion.h) the way the
// reference's tests do (test/test_concat.cc:89-108: build memories, fill through
// data(), create the op, submit(), read dst->data()) and dumps inputs and results as
// raw files; tests/test_dropin.py re-checks them against the CPU oracle.
//   dropin_check <outdir>
#include <cstdio>
#include <cstring>
#include <string>

#include "cli_flags.h"
#include "deepfusion.h"

using namespace deepfusion;

static void dump(const std::string &path, const void *p, size_t bytes) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
  fclose(f);
}

int main(int argc, char **argv) {
  const std::string out = argc > 1 ? argv[1] : ".";
  Lcg g(42);
  // ---- fused conv: N=2, 13x13, 32 -> 32 -> 64, pad 1, u8 out, s32 bias, per-channel scale1 ----
  {
    const int bs = 2, ic = 32, ih = 13, iw = 13, oc = 32, oc1 = 64;
    std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, 3, 3}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> wei1(new memory(memory::nchw_dims{oc1, oc, 1, 1}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> bia(new memory(memory::dims{oc}, memory::format::x, memory::dtype::s32));
    std::unique_ptr<memory> bia1(new memory(memory::dims{oc1}, memory::format::x, memory::dtype::s32));
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc1, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 17);
    std::vector<s8> w0(wei->size()), w1(wei1->size());
    for (auto &v : w0) v = (s8)((int)(g.next() % 21) - 10);
    for (auto &v : w1) v = (s8)((int)(g.next() % 21) - 10);
    reorder_weights(w0.data(), wei);
    reorder_weights(w1.data(), wei1);
    int32_t *b0 = (int32_t *)bia->data(), *b1 = (int32_t *)bia1->data();
    for (int i = 0; i < oc; ++i) b0[i] = (int)(g.next() % 21) - 10;
    for (int i = 0; i < oc1; ++i) b1[i] = (int)(g.next() % 21) - 10;
    std::vector<float> sc1(oc1);
    for (int i = 0; i < oc1; ++i) sc1[i] = 0.02f * (0.5f + (float)i / oc1);
    auto c = conv(src, wei, bia, {1, 1}, {1, 1}, wei1, bia1, dst, true, {1.f / 64}, round_mode::nearest, true, sc1,
                  round_mode::nearest);
    c->submit();
    dump(out + "/fused_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/fused_w0_oihw.bin", w0.data(), w0.size());
    dump(out + "/fused_w1_oihw.bin", w1.data(), w1.size());
    dump(out + "/fused_b0.bin", bia->host_data(), bia->buffer_size());
    dump(out + "/fused_b1.bin", bia1->host_data(), bia1->buffer_size());
    dump(out + "/fused_sc1.bin", sc1.data(), sc1.size() * 4);
    dump(out + "/fused_dst.bin", dst->data(), dst->buffer_size());
    // the caller rewrites the input through data(): the next submit must see it
    for (size_t i = 0; i < src->size(); ++i) ((uint8_t *)src->data())[i] = (uint8_t)(16 - s[i]);
    c->submit();
    dump(out + "/fused_dst2.bin", dst->data(), dst->buffer_size());
    // ... and so must a refill through the pointer `s` fetched once, before the first submit (the
    // reference reads host memory on every submit; the usual inference loop caches the pointer)
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)((i * 7 + 3) % 17);
    c->submit();
    dump(out + "/fused_src3.bin", src->host_data(), src->buffer_size());
    dump(out + "/fused_dst3.bin", dst->host_data(), dst->buffer_size());
  }
  // ---- coherence rules of the asynchronous extension, and op lifetime (two chained unfused convs
  //      a: src -> mid, b: mid -> dst; N=2, 9x9, 32 -> 32 -> 32, u8) ----
  {
    const int bs = 2, ic = 32, ih = 9, iw = 9, oc = 32;
    auto mk_act = [&]() {
      return std::unique_ptr<memory>(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    };
    auto mk_wei = [&]() {
      return std::unique_ptr<memory>(new memory(memory::nchw_dims{oc, ic, 3, 3}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    };
    std::unique_ptr<memory> src = mk_act(), mid = mk_act(), dst = mk_act(), wa = mk_wei(), wb = mk_wei();
    static const std::unique_ptr<memory> none;
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 17);
    std::vector<s8> wa0(wa->size()), wa1(wa->size()), wb0(wb->size());
    for (auto &v : wa0) v = (s8)((int)(g.next() % 21) - 10);
    for (auto &v : wa1) v = (s8)((int)(g.next() % 21) - 10);
    for (auto &v : wb0) v = (s8)((int)(g.next() % 21) - 10);
    reorder_weights(wa0.data(), wa);
    reorder_weights(wb0.data(), wb);
    auto a = conv(src, wa, none, {1, 1}, {1, 1}, mid, true, {1.f / 32}, round_mode::nearest);
    auto b = conv(mid, wb, none, {1, 1}, {1, 1}, dst, true, {1.f / 32}, round_mode::nearest);
    dump(out + "/chain_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/chain_wa0_oihw.bin", wa0.data(), wa0.size());
    dump(out + "/chain_wa1_oihw.bin", wa1.data(), wa1.size());
    dump(out + "/chain_wb_oihw.bin", wb0.data(), wb0.size());
    // Device-resident chaining is the single-device extension; with DEEPFUSION_DEVICES (batch shards, host in ->
    // host out) the same files are produced through synchronous submits.
    const char *dv = getenv("DEEPFUSION_DEVICES");
    const bool sharded = dv && *dv && strcmp(dv, "1") != 0;
    // (1) a synchronous submit packs the weights; the caller then rewrites them through data() and goes on
    //     with submit_async(): the new weights must be used
    a->submit();
    dump(out + "/chain_mid0.bin", mid->host_data(), mid->buffer_size());
    reorder_weights(wa1.data(), wa);  // (writes through wa->data())
    a->submit_async();
    a->wait();
    mid->download();  // (sharded: the shards wrote the host buffer themselves; nothing to fetch)
    dump(out + "/chain_mid1.bin", mid->host_data(), mid->buffer_size());
    // (2) producer asynchronous, consumer synchronous: mid's device copy (weights wa0) is the newer side,
    //     its host bytes are overwritten with garbage that must NOT reach the device
    reorder_weights(wa0.data(), wa);
    if (!sharded) {
      a->submit_async();  // device mid = conv(src, wa0); host mid still holds the wa1 result
      memset(const_cast<void *>(mid->host_data()), 0xEE, mid->buffer_size());
    } else {
      a->submit();
    }
    b->submit();
    dump(out + "/chain_dst_a0.bin", dst->host_data(), dst->buffer_size());
    // (3) the producing op is destroyed before the consumer runs (tensors outlive ops; ops may go)
    if (!sharded) a->submit_async(); else a->submit();
    a.reset();
    b->submit();
    dump(out + "/chain_dst_gone.bin", dst->host_data(), dst->buffer_size());
  }
  // ---- fused conv, N=5 (uneven over 2 or 3 batch shards: DEEPFUSION_DEVICES), s32 out ----
  {
    const int bs = 5, ic = 32, ih = 9, iw = 11, oc = 32, oc1 = 32;
    std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, 3, 3}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> wei1(new memory(memory::nchw_dims{oc1, oc, 1, 1}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc1, ih, iw}, memory::format::nhwc, memory::dtype::s32));
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 256);
    std::vector<s8> w0(wei->size()), w1(wei1->size());
    for (auto &v : w0) v = (s8)((int)(g.next() % 31) - 15);
    for (auto &v : w1) v = (s8)((int)(g.next() % 31) - 15);
    reorder_weights(w0.data(), wei);
    reorder_weights(w1.data(), wei1);
    static const std::unique_ptr<memory> none;
    auto c = conv(src, wei, none, {1, 1}, {1, 1}, wei1, none, dst, true, {1.f / 256}, round_mode::nearest, false, {1.f / 8},
                  round_mode::nearest);
    c->submit();
    dump(out + "/n5_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/n5_w0_oihw.bin", w0.data(), w0.size());
    dump(out + "/n5_w1_oihw.bin", w1.data(), w1.size());
    dump(out + "/n5_dst.bin", dst->data(), dst->buffer_size());
  }
  // ---- unfused conv: N=1, 9x7, 32 -> 48, stride 2, pad 1, s8 out, no relu, round down ----
  {
    const int bs = 1, ic = 32, ih = 9, iw = 7, oc = 48, oh = 5, ow = 4;
    std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, 3, 3}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> bia(new memory(memory::dims{oc}, memory::format::x, memory::dtype::s8));
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc, oh, ow}, memory::format::nhwc, memory::dtype::s8));
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 256);
    std::vector<s8> w0(wei->size());
    for (auto &v : w0) v = (s8)((int)(g.next() % 255) - 127);
    reorder_weights(w0.data(), wei);
    s8 *b0 = (s8 *)bia->data();
    for (int i = 0; i < oc; ++i) b0[i] = (s8)((int)(g.next() % 21) - 10);
    auto c = conv(src, wei, bia, {2, 2}, {1, 1}, dst, false, {1.f / 4096}, round_mode::down);
    c->submit();
    dump(out + "/unfused_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/unfused_w0_oihw.bin", w0.data(), w0.size());
    dump(out + "/unfused_b0.bin", bia->host_data(), bia->buffer_size());
    dump(out + "/unfused_dst.bin", dst->data(), dst->buffer_size());
  }
  // ---- concat + relu: BASELINE.json configs[0] shape (4 x {1,16,8,8} f32) and an s8 case ----
  {
    std::vector<std::unique_ptr<memory>> srcs;
    for (int k = 0; k < 4; ++k) {
      srcs.emplace_back(new memory(memory::nchw_dims{1, 16, 8, 8}, memory::format::nhwc, memory::dtype::f32));
      float *p = (float *)srcs.back()->data();
      for (size_t i = 0; i < srcs.back()->size(); ++i) p[i] = ((int)(g.next() % 2001) - 1000) * 0.01f;
      dump(out + "/concat_f32_src" + std::to_string(k) + ".bin", p, srcs.back()->buffer_size());
    }
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{1, 64, 8, 8}, memory::format::nhwc, memory::dtype::f32));
    auto c = concat(srcs, dst, true);
    c->submit();
    dump(out + "/concat_f32_dst.bin", dst->data(), dst->buffer_size());
  }
  {
    std::vector<std::unique_ptr<memory>> srcs;
    const int chs[3] = {16, 32, 64};
    for (int k = 0; k < 3; ++k) {
      srcs.emplace_back(new memory(memory::nchw_dims{2, chs[k], 3, 3}, memory::format::nhwc, memory::dtype::s8));
      s8 *p = (s8 *)srcs.back()->data();
      for (size_t i = 0; i < srcs.back()->size(); ++i) p[i] = (s8)((int)(g.next() % 256) - 128);
      dump(out + "/concat_s8_src" + std::to_string(k) + ".bin", p, srcs.back()->buffer_size());
    }
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{2, 112, 3, 3}, memory::format::nhwc, memory::dtype::s8));
    auto c = concat(srcs, dst, true);
    c->submit();
    dump(out + "/concat_s8_dst.bin", dst->data(), dst->buffer_size());
  }
  // ---- conv + relu + 2x2/2 max pool (roadmap op; first VGG-style shape of test_conv_relu_pooling.cc:322-324
  //      scaled to 32 channels, odd image size so the pool windows hang over the edge) ----
  {
    const int bs = 3, ic = 32, ih = 11, iw = 9, oc = 48, ph = 6, pw = 5;
    std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, 3, 3}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> bia(new memory(memory::dims{oc}, memory::format::x, memory::dtype::s32));
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc, ph, pw}, memory::format::nhwc, memory::dtype::u8));
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 64);
    std::vector<s8> w0(wei->size());
    for (auto &v : w0) v = (s8)((int)(g.next() % 21) - 10);
    reorder_weights(w0.data(), wei);
    int32_t *b0 = (int32_t *)bia->data();
    for (int i = 0; i < oc; ++i) b0[i] = (int)(g.next() % 201) - 100;
    auto c = conv_relu_pool(src, wei, bia, {1, 1}, {1, 1}, {2, 2}, {2, 2}, {0, 0}, dst, true, {1.f / 32});
    c->submit();
    dump(out + "/pool_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/pool_w0_oihw.bin", w0.data(), w0.size());
    dump(out + "/pool_b0.bin", bia->host_data(), bia->buffer_size());
    dump(out + "/pool_dst.bin", dst->data(), dst->buffer_size());
  }
  // ---- the same op on an even-sized conv output, 64 channels: the pooling is fused into the conv kernel ----
  {
    const int bs = 2, ic = 64, ih = 12, iw = 40, oc = 64, ph = 6, pw = 20;
    std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, 3, 3}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> bia(new memory(memory::dims{oc}, memory::format::x, memory::dtype::s32));
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc, ph, pw}, memory::format::nhwc, memory::dtype::u8));
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 64);
    std::vector<s8> w0(wei->size());
    for (auto &v : w0) v = (s8)((int)(g.next() % 21) - 10);
    reorder_weights(w0.data(), wei);
    int32_t *b0 = (int32_t *)bia->data();
    for (int i = 0; i < oc; ++i) b0[i] = (int)(g.next() % 201) - 100;
    auto c = conv_relu_pool(src, wei, bia, {1, 1}, {1, 1}, {2, 2}, {2, 2}, {0, 0}, dst, true, {1.f / 64});
    c->submit();
    dump(out + "/poolf_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/poolf_w0_oihw.bin", w0.data(), w0.size());
    dump(out + "/poolf_b0.bin", bia->host_data(), bia->buffer_size());
    dump(out + "/poolf_dst.bin", dst->data(), dst->buffer_size());
  }
  // ---- 1x1 conv + relu + 7x7 global average (exclude padding): the ResNet-head instance of
  //      test_conv_relu_pooling.cc:334-335, scaled to 64 -> 128 channels ----
  {
    const int bs = 3, ic = 64, ih = 7, iw = 7, oc = 128;
    std::unique_ptr<memory> src(new memory(memory::nchw_dims{bs, ic, ih, iw}, memory::format::nhwc, memory::dtype::u8));
    std::unique_ptr<memory> wei(new memory(memory::nchw_dims{oc, ic, 1, 1}, memory::format::OIhw4i16o4i, memory::dtype::s8));
    std::unique_ptr<memory> bia(new memory(memory::dims{oc}, memory::format::x, memory::dtype::s32));
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{bs, oc, 1, 1}, memory::format::nhwc, memory::dtype::u8));
    uint8_t *s = (uint8_t *)src->data();
    for (size_t i = 0; i < src->size(); ++i) s[i] = (uint8_t)(g.next() % 200);
    std::vector<s8> w0(wei->size());
    for (auto &v : w0) v = (s8)((int)(g.next() % 15) - 7);
    reorder_weights(w0.data(), wei);
    int32_t *b0 = (int32_t *)bia->data();
    for (int i = 0; i < oc; ++i) b0[i] = (int)(g.next() % 2001) - 1000;
    auto c = conv_relu_pool(src, wei, bia, {1, 1}, {0, 0}, {7, 7}, {7, 7}, {0, 0}, dst, true, {1.f / 64},
                            round_mode::nearest, pool_algo::avg_exclude_padding);
    c->submit();
    dump(out + "/avg_src.bin", src->host_data(), src->buffer_size());
    dump(out + "/avg_w0_oihw.bin", w0.data(), w0.size());
    dump(out + "/avg_b0.bin", bia->host_data(), bia->buffer_size());
    dump(out + "/avg_dst.bin", dst->data(), dst->buffer_size());
  }
  // ---- eltwise sum + relu of three s8 tensors (roadmap op) ----
  {
    std::vector<std::unique_ptr<memory>> srcs;
    for (int k = 0; k < 3; ++k) {
      srcs.emplace_back(new memory(memory::nchw_dims{2, 24, 5, 7}, memory::format::nhwc, memory::dtype::s8));
      s8 *p = (s8 *)srcs.back()->data();
      for (size_t i = 0; i < srcs.back()->size(); ++i) p[i] = (s8)((int)(g.next() % 256) - 128);
      dump(out + "/elt_s8_src" + std::to_string(k) + ".bin", p, srcs.back()->buffer_size());
    }
    std::unique_ptr<memory> dst(new memory(memory::nchw_dims{2, 24, 5, 7}, memory::format::nhwc, memory::dtype::s8));
    auto c = eltwise_sum(srcs, dst, true);
    c->submit();
    dump(out + "/elt_s8_dst.bin", dst->data(), dst->buffer_size());
  }
  printf("dropin_check wrote results to %s\n", out.c_str());
  return 0;
}
