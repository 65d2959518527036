"""CPU-only: pins the C oracle (scalar + AVX-512 witness) against the independent
float64/numpy formulation and against the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

import cases as C
import refmath


def run_oracle(orc, case, data, impl):
    wb = orc.reorder_oihw_to_blocked(data["w0"])
    w1b = orc.reorder_oihw_to_blocked(data["w1"]) if case.oc1x1 else None
    return orc.conv(data["src"], wb, data["w0"].shape, case.stride, case.pad, case.dst_dt,
                    data["scales0"], bia0=data["bia0"], wei1_blk=w1b, oc1x1=case.oc1x1,
                    scales1=data["scales1"], bia1=data["bia1"], relu0=case.relu0,
                    relu1=case.relu1, rm0=case.rm0, rm1=case.rm1, impl=impl)


def assert_same(a, b):
    assert a.dtype == b.dtype and a.shape == b.shape
    if a.dtype == np.float32:
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    else:
        assert np.array_equal(a, b)


ORACLE_CASES = (C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL) + C.option_sweep(C.SMALL64)
                + C.dtype_matrix(C.REF_SHAPES[0]) + [C.REF_SHAPES[1], C.CONFIG2]
                + [C.unfused(c) for c in C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL)]
                + [C.ConvCase("s2", 1, 16, 11, 9, 48, 80, stride=(2, 2)),
                   C.ConvCase("k5", 1, 16, 9, 9, 16, 16, k=(5, 5), pad=(2, 2)),
                   C.ConvCase("k1", 2, 32, 5, 5, 32, 0, k=(1, 1), pad=(0, 0), dst_dt=C.S32)])


@pytest.mark.parametrize("case", ORACLE_CASES, ids=lambda c: c.ident())
def test_scalar_oracle_vs_independent(oracle, case):
    data = C.generate(case)
    got = run_oracle(oracle, case, data, "scalar")
    ref = refmath.conv_ref(case, data)
    assert_same(got, ref)


@pytest.mark.parametrize("case", ORACLE_CASES + [C.REF_SHAPES[2], C.CONFIG3_SMALL],
                         ids=lambda c: c.ident())
def test_avx512_witness_vs_independent(oracle, case):
    if not oracle.have_avx512_vnni():
        pytest.skip("host lacks AVX-512 VNNI")
    data = C.generate(case)
    got = run_oracle(oracle, case, data, "avx512")
    ref = refmath.conv_ref(case, data)
    assert_same(got, ref)


def test_cvt_edge_cases(oracle):
    """NaN / overflow follow vcvtps2dq (0x80000000), -0.0 survives vmaxps."""
    case = C.ConvCase("edge", 1, 16, 3, 3, 16, 16, dst_dt=C.S32, bia0_dt=C.UNDEF,
                      bia1_dt=C.UNDEF, relu1=False, wide=True)
    data = C.generate(case)
    data["scales1"] = np.array([3.0e38], dtype=np.float32)      # overflow -> indefinite
    for impl in ("scalar", "avx512"):
        if impl == "avx512" and not oracle.have_avx512_vnni():
            continue
        got = run_oracle(oracle, case, data, impl)
        assert_same(got, refmath.conv_ref(case, data))
        assert (got == -2147483648).any()
    case = C.ConvCase("negzero", 1, 16, 3, 3, 16, 16, dst_dt=C.F32, bia0_dt=C.UNDEF,
                      bia1_dt=C.UNDEF, relu1=True)
    data = C.generate(case)
    data["scales1"] = np.array([-0.0], dtype=np.float32)         # acc * -0.0 = -0.0 for acc >= 0
    for impl in ("scalar", "avx512"):
        if impl == "avx512" and not oracle.have_avx512_vnni():
            continue
        got = run_oracle(oracle, case, data, impl)
        assert_same(got, refmath.conv_ref(case, data))
        assert (got.view(np.uint32) == 0x80000000).any()


def test_blocked_layout_known_answer(oracle):
    """element (o,i,h,w) sits at [o/16][i/16][kh][kw][(i%16)/4][o%16][i%4]
    (reference src/jit_conv_kernel.cc:333-338)."""
    O, I, KH, KW = 32, 32, 3, 3
    w = np.arange(O * I * KH * KW, dtype=np.int64).reshape(O, I, KH, KW)
    w8 = (w % 251 - 125).astype(np.int8)
    blk = oracle.reorder_oihw_to_blocked(w8)
    b = blk.reshape(O // 16, I // 16, KH, KW, 4, 16, 4)
    for (o, i, h, x) in [(0, 0, 0, 0), (17, 5, 2, 1), (31, 31, 2, 2), (16, 15, 1, 0), (3, 20, 0, 2)]:
        assert b[o // 16, i // 16, h, x, (i % 16) // 4, o % 16, i % 4] == w8[o, i, h, x]


# concat: the 9 shapes of reference test/test_concat.cc:122-145
CONCAT_SHAPES = [
    ([(2, 64, 1, 1), (2, 96, 1, 1)]), ([(2, 64, 4, 4), (2, 32, 4, 4)]),
    ([(2, 16, 8, 8), (2, 32, 8, 8)]), ([(2, 32, 9, 9), (2, 96, 9, 9)]),
    ([(2, 16, 3, 3), (2, 32, 3, 3), (2, 64, 3, 3)]), ([(2, 256, 16, 16), (2, 256, 16, 16)]),
    ([(4, 128, 14, 14), (4, 256, 14, 14)]),
]
CONCAT_NARROW = [([(2, 4, 4, 4), (2, 8, 4, 4)]), ([(2, 16, 4, 4), (2, 8, 4, 4)])]


def concat_inputs(shapes, np_dt, wide, seed=7):
    rng = np.random.default_rng(seed)
    srcs = []
    for (n, c, h, w) in shapes:
        sz = (n, h, w, c)
        if np_dt == np.float32:
            i = np.arange(int(np.prod(sz)))
            v = (1.0 + 0.01 * np.sin((i % 37).astype(np.float32))).astype(np.float32)
            if wide:
                v = (v - 1.0) * 100.0
            srcs.append(v.reshape(sz).astype(np.float32))
        elif np_dt == np.uint8:
            srcs.append(rng.integers(0, 256 if wide else 17, sz).astype(np.uint8))
        elif np_dt == np.int8:
            srcs.append(rng.integers(-128 if wide else -10, 128 if wide else 11, sz).astype(np.int8))
        else:
            lim = 2 ** 31 - 1 if wide else 10
            srcs.append(rng.integers(-lim, lim + 1, sz).astype(np.int32))
    return srcs


@pytest.mark.parametrize("np_dt", [np.float32, np.int32, np.int8, np.uint8])
@pytest.mark.parametrize("wide", [False, True])
def test_concat_oracle(oracle, np_dt, wide):
    shapes = CONCAT_SHAPES + (CONCAT_NARROW if np_dt in (np.float32, np.int32) else [])
    for sh in shapes:
        srcs = concat_inputs(sh, np_dt, wide)
        for relu in (True, False):
            got = oracle.concat(srcs, relu)
            ref = np.concatenate(srcs, axis=3)
            if relu and np_dt != np.uint8:
                ref = np.maximum(ref, np_dt(0))
            assert np.array_equal(got, ref)


def test_golden_vectors(oracle):
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "conv_*.npz")))
    assert files, "golden fixtures missing (run tests/golden/make_golden.py)"
    for f in files:
        z = np.load(f, allow_pickle=False)
        case = C.ConvCase(**{k: (tuple(v.tolist()) if v.ndim else v.item())
                             for k, v in ((k[5:], z[k]) for k in z.files if k.startswith("case_"))})
        data = dict(src=z["src"], w0=z["w0"], w1=z["w1"] if "w1" in z.files else None,
                    bia0=z["bia0"] if "bia0" in z.files else None,
                    bia1=z["bia1"] if "bia1" in z.files else None,
                    scales0=z["scales0"], scales1=z["scales1"])
        for impl in ("scalar", "avx512"):
            if impl == "avx512" and not oracle.have_avx512_vnni():
                continue
            assert_same(run_oracle(oracle, case, data, impl), z["dst"])


# ---- the reference's roadmap ops (README.md:64-65); shapes after test/test_conv_relu_pooling.cc:313-338
# (2x2 stride-2 max pooling behind VGG-style convs, 7x7 global window), plus windows that hang over the
# edges, padding, odd channel counts.  No reference implementation or vectors exist: parity unpinned --
# the oracle is checked against an independent numpy formulation only. ----
POOL_CASES = [
    # (n, h, w, c), kernel, stride, pad, (oh, ow)
    ((1, 2, 2, 16), (2, 2), (2, 2), (0, 0), (1, 1)),
    ((1, 224, 224, 16), (2, 2), (2, 2), (0, 0), (112, 112)),
    ((2, 56, 56, 64), (2, 2), (2, 2), (0, 0), (28, 28)),
    ((3, 7, 7, 48), (7, 7), (7, 7), (0, 0), (1, 1)),
    ((2, 11, 9, 32), (2, 2), (2, 2), (0, 0), (6, 5)),      # windows over the bottom / right edge
    ((2, 13, 12, 20), (3, 3), (2, 2), (1, 1), (7, 6)),     # ResNet-stem style 3x3/2 pad 1
    ((1, 5, 6, 3), (3, 2), (1, 2), (1, 0), (5, 3)),        # odd channel count: scalar path
    ((4, 9, 9, 5), (3, 3), (3, 3), (0, 0), (3, 3)),
]


def pool_input(shape, np_dt, seed=0):
    rng = np.random.default_rng(seed)
    if np_dt == np.float32:
        return (rng.standard_normal(shape) * 100).astype(np.float32)
    info = np.iinfo(np_dt)
    return rng.integers(info.min, int(info.max) + 1, shape).astype(np_dt)


@pytest.mark.parametrize("np_dt", [np.uint8, np.int8, np.int32, np.float32])
@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "x".join(map(str, c[0])))
def test_maxpool_oracle_vs_independent(oracle, case, np_dt):
    shape, k, s, p, o = case
    if shape[1] > 100 and np_dt != np.uint8:
        pytest.skip("large shape once")
    x = pool_input(shape, np_dt)
    assert np.array_equal(oracle.maxpool(x, k, s, p, o), refmath.maxpool(x, k, s, p, o))


@pytest.mark.parametrize("include_padding", [True, False])
@pytest.mark.parametrize("np_dt", [np.uint8, np.int8, np.int32, np.float32])
@pytest.mark.parametrize("case", [c for c in POOL_CASES if c[0][1] <= 100], ids=lambda c: "x".join(map(str, c[0])))
def test_avgpool_oracle_vs_independent(oracle, case, np_dt, include_padding):
    """the reference test's pooling_avg_include_padding / _exclude_padding flags (test_conv_relu_pooling.cc:189-193;
    one instance: the 7x7 global average behind ResNet's last block, :334-335).  Parity unpinned."""
    shape, k, s, p, o = case
    x = pool_input(shape, np_dt, seed=7)
    a = oracle.avgpool(x, k, s, p, o, include_padding)
    b = refmath.avgpool(x, k, s, p, o, include_padding)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_maxpool_oracle_rejects_window_in_padding(oracle):
    with pytest.raises(RuntimeError):
        oracle.maxpool(np.zeros((1, 4, 4, 16), np.uint8), (2, 2), (2, 2), (0, 0), (4, 4))


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("np_dt", [np.uint8, np.int8, np.int32, np.float32])
def test_eltwise_sum_oracle_vs_independent(oracle, np_dt, relu):
    for shape, n in (((2, 5, 7, 24), 3), ((1, 1, 1, 1), 2), ((3, 9, 4, 17), 8)):
        xs = [pool_input(shape, np_dt, seed=10 + k) for k in range(n)]
        a, b = oracle.eltwise_sum(xs, relu), refmath.eltwise_sum(xs, relu)
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    if np_dt != np.float32:  # saturation at both ends
        info = np.iinfo(np_dt)
        xs = [np.full((4, 4), info.max, np_dt), np.full((4, 4), info.max, np_dt)]
        assert (oracle.eltwise_sum(xs, relu) == info.max).all()
        xs = [np.full((4, 4), info.min, np_dt), np.full((4, 4), info.min, np_dt)]
        assert (oracle.eltwise_sum(xs, relu) == (0 if relu else info.min)).all()
