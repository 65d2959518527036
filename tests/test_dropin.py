"""GPU: the drop-in C++ API (include/deepfusion.h, libdeepfusion.so) used the way the
reference's tests use it (memories filled through data(), op created, submit(), dst
read through data()), checked bit-for-bit against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import cases as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "deep-fusion_amd", "tools")


def _load(d, name, dtype, shape=None):
    a = np.fromfile(os.path.join(d, name), dtype=dtype)
    return a.reshape(shape) if shape else a


def test_dropin_cpp_api(oracle, tmp_path):
    exe = os.path.join(TOOLS, "dropin_check")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    subprocess.check_call([exe, str(tmp_path)])
    d = str(tmp_path)
    # fused conv
    src = _load(d, "fused_src.bin", np.uint8, (2, 13, 13, 32))
    w0 = _load(d, "fused_w0_oihw.bin", np.int8, (32, 32, 3, 3))
    w1 = _load(d, "fused_w1_oihw.bin", np.int8, (64, 32, 1, 1))
    b0 = _load(d, "fused_b0.bin", np.int32)
    b1 = _load(d, "fused_b1.bin", np.int32)
    sc1 = _load(d, "fused_sc1.bin", np.float32)
    sc0 = np.array([1.0 / 64], dtype=np.float32)

    def ref_fused(s):
        return oracle.conv(s, oracle.reorder_oihw_to_blocked(w0), w0.shape, (1, 1), (1, 1), C.U8, sc0,
                           bia0=b0, wei1_blk=oracle.reorder_oihw_to_blocked(w1), oc1x1=64, scales1=sc1,
                           bia1=b1, relu0=True, relu1=True)
    got = _load(d, "fused_dst.bin", np.uint8, (2, 13, 13, 64))
    assert np.array_equal(got, ref_fused(src))
    got2 = _load(d, "fused_dst2.bin", np.uint8, (2, 13, 13, 64))       # input rewritten via data()
    assert np.array_equal(got2, ref_fused((16 - src.astype(np.int32)).astype(np.uint8)))
    assert not np.array_equal(got, got2)
    src3 = _load(d, "fused_src3.bin", np.uint8, (2, 13, 13, 32))       # refilled through a cached pointer
    assert np.array_equal(_load(d, "fused_dst3.bin", np.uint8, (2, 13, 13, 64)), ref_fused(src3))
    assert not np.array_equal(src3, src)
    # coherence of the asynchronous extension + op lifetime: two chained unfused convs
    csrc = _load(d, "chain_src.bin", np.uint8, (2, 9, 9, 32))
    wa0, wa1, wbb = (_load(d, "chain_%s_oihw.bin" % n, np.int8, (32, 32, 3, 3)) for n in ("wa0", "wa1", "wb"))
    sc = np.array([1.0 / 32], dtype=np.float32)

    def ref_plain(x, w):
        return oracle.conv(x, oracle.reorder_oihw_to_blocked(w), w.shape, (1, 1), (1, 1), C.U8, sc, relu0=True)
    mid0, mid1 = ref_plain(csrc, wa0), ref_plain(csrc, wa1)
    assert not np.array_equal(mid0, mid1)
    assert np.array_equal(_load(d, "chain_mid0.bin", np.uint8, mid0.shape), mid0)
    assert np.array_equal(_load(d, "chain_mid1.bin", np.uint8, mid1.shape), mid1), \
        "submit_async() after a synchronous submit ignored weights rewritten through data()"
    want = ref_plain(mid0, wbb)
    assert np.array_equal(_load(d, "chain_dst_a0.bin", np.uint8, want.shape), want), \
        "a synchronous consumer uploaded stale host bytes over an asynchronous producer's device result"
    assert np.array_equal(_load(d, "chain_dst_gone.bin", np.uint8, want.shape), want), \
        "consumer after the producing op was destroyed"
    # fused conv N=5, s32 out, no bias
    s5 = _load(d, "n5_src.bin", np.uint8, (5, 9, 11, 32))
    w50 = _load(d, "n5_w0_oihw.bin", np.int8, (32, 32, 3, 3))
    w51 = _load(d, "n5_w1_oihw.bin", np.int8, (32, 32, 1, 1))
    ref5 = oracle.conv(s5, oracle.reorder_oihw_to_blocked(w50), w50.shape, (1, 1), (1, 1), C.S32,
                       np.array([1.0 / 256], dtype=np.float32), wei1_blk=oracle.reorder_oihw_to_blocked(w51),
                       oc1x1=32, scales1=np.array([1.0 / 8], dtype=np.float32), relu0=True, relu1=False)
    assert np.array_equal(_load(d, "n5_dst.bin", np.int32, ref5.shape), ref5)
    # unfused conv, stride 2, s8 out, round down
    src = _load(d, "unfused_src.bin", np.uint8, (1, 9, 7, 32))
    w0 = _load(d, "unfused_w0_oihw.bin", np.int8, (48, 32, 3, 3))
    b0 = _load(d, "unfused_b0.bin", np.int8)
    ref = oracle.conv(src, oracle.reorder_oihw_to_blocked(w0), w0.shape, (2, 2), (1, 1), C.S8,
                      np.array([1.0 / 4096], dtype=np.float32), bia0=b0, relu0=False, rm0=1)
    assert np.array_equal(_load(d, "unfused_dst.bin", np.int8, ref.shape), ref)
    # concat + relu
    srcs = [_load(d, "concat_f32_src%d.bin" % k, np.float32, (1, 8, 8, 16)) for k in range(4)]
    ref = oracle.concat(srcs, True)
    assert np.array_equal(_load(d, "concat_f32_dst.bin", np.float32, ref.shape).view(np.uint32), ref.view(np.uint32))
    srcs = [_load(d, "concat_s8_src%d.bin" % k, np.int8, (2, 3, 3, c)) for k, c in enumerate((16, 32, 64))]
    ref = oracle.concat(srcs, True)
    assert np.array_equal(_load(d, "concat_s8_dst.bin", np.int8, ref.shape), ref)
    # roadmap ops: conv + relu + 2x2/2 max pool (windows hang over the bottom / right edge), eltwise sum + relu
    src = _load(d, "pool_src.bin", np.uint8, (3, 11, 9, 32))
    w0 = _load(d, "pool_w0_oihw.bin", np.int8, (48, 32, 3, 3))
    b0 = _load(d, "pool_b0.bin", np.int32)
    mid = oracle.conv(src, oracle.reorder_oihw_to_blocked(w0), w0.shape, (1, 1), (1, 1), C.U8,
                      np.array([1.0 / 32], dtype=np.float32), bia0=b0, relu0=True)
    ref = oracle.maxpool(mid, (2, 2), (2, 2), (0, 0), (6, 5))
    assert np.array_equal(_load(d, "pool_dst.bin", np.uint8, ref.shape), ref)
    src = _load(d, "poolf_src.bin", np.uint8, (2, 12, 40, 64))    # even-sized: pooling fused into the conv kernel
    w0 = _load(d, "poolf_w0_oihw.bin", np.int8, (64, 64, 3, 3))
    b0 = _load(d, "poolf_b0.bin", np.int32)
    mid = oracle.conv(src, oracle.reorder_oihw_to_blocked(w0), w0.shape, (1, 1), (1, 1), C.U8,
                      np.array([1.0 / 64], dtype=np.float32), bia0=b0, relu0=True)
    ref = oracle.maxpool(mid, (2, 2), (2, 2), (0, 0), (6, 20))
    assert np.array_equal(_load(d, "poolf_dst.bin", np.uint8, ref.shape), ref)
    src = _load(d, "avg_src.bin", np.uint8, (3, 7, 7, 64))          # 1x1 conv + relu + 7x7 global average
    w0 = _load(d, "avg_w0_oihw.bin", np.int8, (128, 64, 1, 1))
    b0 = _load(d, "avg_b0.bin", np.int32)
    mid = oracle.conv(src, oracle.reorder_oihw_to_blocked(w0), w0.shape, (1, 1), (0, 0), C.U8,
                      np.array([1.0 / 64], dtype=np.float32), bia0=b0, relu0=True)
    ref = oracle.avgpool(mid, (7, 7), (7, 7), (0, 0), (1, 1), False)
    assert np.array_equal(_load(d, "avg_dst.bin", np.uint8, ref.shape), ref)
    srcs = [_load(d, "elt_s8_src%d.bin" % k, np.int8, (2, 5, 7, 24)) for k in range(3)]
    ref = oracle.eltwise_sum(srcs, True)
    assert np.array_equal(_load(d, "elt_s8_dst.bin", np.int8, ref.shape), ref)


def test_bench_tools_run():
    """the reference's bench flag names are accepted (bench_concat.cc:22-29, bench_conv.cc:22-37)."""
    out = subprocess.check_output([os.path.join(TOOLS, "bench_concat"), "-n", "1", "-c", "16,16,16,16", "-h", "8",
                                   "-w", "8", "-dtype", "f32", "-post_relu", "-burning_iter", "2", "-iter", "3"])
    assert b"DeepFusion Concat avg time" in out
    out = subprocess.check_output([os.path.join(TOOLS, "bench_conv"), "-bs", "2", "-ih", "28", "-iw", "28", "-kh", "3",
                                   "-kw", "3", "-sh", "1", "-sw", "1", "-ph", "1", "-pw", "1", "-ic", "32", "-oc", "32",
                                   "-oc1x1", "64", "-dtype", "u8", "-burning_iter", "2", "-iter", "3"])
    assert b"DeepFusion Conv avg time" in out
    # -cold_cache: the reference's cold-cache protocol (CMakeLists.txt:60-61, test/test_utils.cc:23-45)
    for tool, args in (("bench_concat", ["-n", "1", "-c", "16,16", "-h", "8", "-w", "8", "-dtype", "s8"]),
                       ("bench_conv", ["-bs", "2", "-ih", "28", "-iw", "28", "-ic", "32", "-oc", "32", "-oc1x1", "64",
                                       "-dtype", "u8"])):
        out = subprocess.check_output([os.path.join(TOOLS, tool)] + args + ["-burning_iter", "1", "-iter", "2", "-cold_cache"])
        assert b"COLD caches" in out and b"warm caches" in out, out


def test_init_failure_exits_like_reference(tmp_path):
    """construction failure -> message + exit(EXIT_FAILURE) (reference op_conv.h:66-68, log.h:38-42)."""
    r = subprocess.run([os.path.join(TOOLS, "bench_conv"), "-bs", "1", "-ih", "8", "-iw", "8", "-ic", "24", "-oc", "32",
                        "-oc1x1", "32", "-burning_iter", "0", "-iter", "1"], capture_output=True)
    assert r.returncode == 1 and b"[deepfusion]" in r.stderr and b"failed" in r.stderr


@pytest.mark.parametrize("shards", ["2", "3", "all"])
def test_dropin_multi_device_same_bytes(tmp_path, shards):
    """DEEPFUSION_DEVICES=n: the drop-in layer splits each op's batch over n shards (device i % count,
    own handle / stream / buffers per shard).  Every output file must equal the single-device run's
    byte for byte -- on a one-GPU box the shards share the device, on an N-GPU node they spread."""
    exe = os.path.join(TOOLS, "dropin_check")
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir()
    many.mkdir()
    env = {k: v for k, v in os.environ.items() if k != "DEEPFUSION_DEVICES"}
    subprocess.check_call([exe, str(one)], env=env)
    subprocess.check_call([exe, str(many)], env=dict(env, DEEPFUSION_DEVICES=shards))
    names = sorted(os.listdir(str(one)))
    assert names == sorted(os.listdir(str(many))) and any("dst" in n for n in names)
    for n in names:
        assert (one / n).read_bytes() == (many / n).read_bytes(), n
