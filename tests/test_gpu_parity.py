"""GPU parity tests (run with -m gpu on the MI355X box).  Every check goes
through the C ABI of include/dfx.h and compares with the CPU oracle on the same
seeded inputs.  Bars: bit-exact for u8/s8/s32 outputs; f32 outputs are also
compared bit-for-bit (tolerance 0 ulp; north_star allows 1)."""
import glob
import os
from dataclasses import replace

import numpy as np
import pytest

import cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import hipref
    return hipref


FUSED_CASES = (C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL) + C.option_sweep(C.SMALL64)
               + C.dtype_matrix(C.REF_SHAPES[0]) + C.dtype_matrix(C.REF_SHAPES[1])
               + [C.REF_SHAPES[2], C.CONFIG2, replace(C.CONFIG2, dst_dt=C.S32, wide=True),
                  C.CONFIG3_SMALL, replace(C.CONFIG3_SMALL, dst_dt=C.U8, wide=True),
                  replace(C.CONFIG3_SMALL, dst_dt=C.F32, per_channel1=True, relu1=False),
                  replace(C.CONFIG3_SMALL, dst_dt=C.S8, relu1=False, wide=True, rm0=1, rm1=1),
                  C.ConvCase("w96", 1, 64, 6, 96, 64, 128, dst_dt=C.S32),
                  C.ConvCase("w40p0", 2, 32, 7, 40, 64, 96, pad=(0, 0), dst_dt=C.S32),
                  C.ConvCase("p10", 1, 32, 9, 9, 32, 32, pad=(1, 0)),
                  C.ConvCase("one", 1, 32, 3, 3, 32, 32, pad=(0, 0))])


@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: c.ident())
def test_fused_auto_variant(hip, oracle, case):
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant == hip.dfa.VARIANT_MFMA_FUSED, info.kernel_name
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode())


UNFUSED_MFMA_CASES = ([C.unfused(c) for c in C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL) + C.option_sweep(C.SMALL64)]
                      + [C.unfused(C.CONFIG2), C.unfused(C.REF_SHAPES[2]), C.unfused(replace(C.CONFIG3_SMALL, dst_dt=C.U8)),
                         C.unfused(replace(C.CONFIG3_SMALL, dst_dt=C.F32, relu0=False, per_channel0=True)),
                         C.unfused(C.ConvCase("w96", 1, 64, 6, 96, 32, 0, dst_dt=C.S32, pad=(0, 1)))])


@pytest.mark.parametrize("case", UNFUSED_MFMA_CASES, ids=lambda c: c.ident())
def test_unfused_mfma_variant(hip, oracle, case):
    """the unfused conv() overload on the int8-MFMA kernel (D0[px][oc] orientation)."""
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant == hip.dfa.VARIANT_MFMA_CONV, info.kernel_name
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode())


GENERIC_CASES = (C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL)
                 + [C.unfused(c) for c in C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL64)]
                 + [C.CONFIG2, C.REF_SHAPES[0],
                    C.ConvCase("s2", 1, 16, 11, 9, 48, 80, stride=(2, 2)),
                    C.ConvCase("k5", 1, 16, 9, 9, 16, 16, k=(5, 5), pad=(2, 2)),
                    C.ConvCase("k1", 2, 32, 5, 5, 32, 0, k=(1, 1), pad=(0, 0), dst_dt=C.S32),
                    C.ConvCase("ic128", 1, 128, 6, 6, 80, 48, dst_dt=C.S32, wide=True)])


@pytest.mark.parametrize("case", GENERIC_CASES, ids=lambda c: c.ident())
def test_generic_variant(hip, oracle, case):
    data = C.generate(case)
    got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_GENERIC)
    assert info.variant == hip.dfa.VARIANT_GENERIC
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode())


COOP_CASES = [
    C.ConvCase("coop_u8", 64, 64, 56, 56, 64, 256, dst_dt=C.U8, wide=True),
    C.ConvCase("coop_s8x", 64, 32, 40, 72, 64, 96, dst_dt=C.S8, relu1=False, rm0=1, rm1=1, per_channel0=True),
    C.unfused(C.ConvCase("coop_unf", 64, 64, 56, 56, 64, 0, dst_dt=C.S32, relu0=False)),
    C.ConvCase("coop_p0", 96, 32, 30, 62, 32, 64, pad=(0, 0), dst_dt=C.F32),
]


@pytest.mark.parametrize("case", COOP_CASES, ids=lambda c: c.ident())
def test_resident_kernel_cooperative_first_tile(hip, oracle, case):
    """enough units per team that the first ones are owned statically: the compute waves stage
    each team's first tile together with the weights (conv_mfma.cuh, coop0)."""
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant in (hip.dfa.VARIANT_MFMA_FUSED, hip.dfa.VARIANT_MFMA_CONV), info.kernel_name
    units = case.bs * -(-case.oh // max(info.rows_per_unit, 1))   # (lower bound: x 1 column unit)
    assert units >= 2 * 2 * info.grid, "shape too small to reach the statically owned first units"
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode())


MODE_CASES = [C.SMALL64, replace(C.SMALL64, dst_dt=C.S32), replace(C.SMALL, dst_dt=C.S8, relu1=False),
              replace(C.CONFIG3_SMALL, dst_dt=C.U8), C.CONFIG3_SMALL, replace(C.CONFIG3_SMALL, dst_dt=C.F32, relu1=False),
              replace(C.SMALL64, bia0_dt=C.F32, bia1_dt=C.F32), replace(C.SMALL64, bia0_dt=C.UNDEF, bia1_dt=C.U8, dst_dt=C.S32),
              C.unfused(replace(C.CONFIG3_SMALL, dst_dt=C.U8)), C.unfused(replace(C.SMALL64, dst_dt=C.S32)),
              C.unfused(replace(C.SMALL, dst_dt=C.F32, relu0=False))]


@pytest.mark.parametrize("switch", ["DFX_NO_MAGIC", "DFX_NO_FAST"])
@pytest.mark.parametrize("case", MODE_CASES, ids=lambda c: c.ident())
def test_resident_kernel_requant_modes(hip, oracle, tuning, case, switch):
    """the resident-weight kernel's three requant modes (conv_mfma.cuh header): the default
    picks "magic" for these reference-range cases; the switches force "fast" (v_cvt_f32_i32
    chain) and "exact" (x86 instruction chain) on the same inputs."""
    tuning.setenv(switch, "1")
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant in (hip.dfa.VARIANT_MFMA_FUSED, hip.dfa.VARIANT_MFMA_CONV), info.kernel_name
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode() + " " + switch)


SCHED_CASES = [
    # >= 3 rounds of units per loader at 2-row units: static split, lazy queue, eager queue all apply
    C.ConvCase("sched_s32", 80, 64, 56, 56, 64, 256, dst_dt=C.S32),
    C.ConvCase("sched_u8", 80, 64, 56, 56, 64, 256, dst_dt=C.U8),
    # many units per loader (three static rounds, then the queue), partial last round
    C.ConvCase("sched_f32", 9, 32, 120, 200, 64, 0, dst_dt=C.F32),
    # store-bound with an odd number of output rows: the bottom unit of an image has one row, its second HALF none
    C.ConvCase("sched_odd", 80, 64, 55, 56, 64, 256, dst_dt=C.S32),
]


@pytest.mark.parametrize("case", SCHED_CASES, ids=lambda c: c.ident())
def test_resident_kernel_unit_scheduling(hip, oracle, tuning, case):
    """how units reach the loaders (conv_mfma.cuh: static stream-major split, lazy queue draws for
    store-bound ops, eager queue, the last units handed out as halves) must not change a byte: every mode is compared with the default
    run, and the default run's first images with the oracle."""
    data = C.generate(case)
    ref_run, info = hip.hip_conv(case, data)
    assert info.variant in (hip.dfa.VARIANT_MFMA_FUSED, hip.dfa.VARIANT_MFMA_CONV), info.kernel_name
    n = min(case.bs, 6)
    sub = dict(data, src=data["src"][:n])
    hip.assert_bit_equal(ref_run[:n], hip.oracle_conv(oracle, replace(case, bs=n), sub), info.kernel_name.decode())
    for key, val in (("DFX_NO_LAZY", "1"), ("DFX_STATIC_ROUNDS", "1"), ("DFX_STATIC_ROUNDS", "2"),
                     ("DFX_STATIC_ROUNDS", "99"), ("DFX_FORCE_GEOM", "4,56" if case.iw == 56 else "3,32"),
                     ("DFX_HALF_UNITS", "0"), ("DFX_HALF_UNITS", "77"), ("DFX_HALF_UNITS", "1000000")):
        tuning.setenv(key, val)
        got, _ = hip.hip_conv(case, data)
        tuning.undo()
        assert np.array_equal(got, ref_run), "%s=%s changes the result" % (key, val)


# streamed-weight MFMA variant (conv_stream.cuh): general shapes -- SURVEY.md 8(f) rank 3
STREAM_SHAPES = [
    C.ConvCase("s2", 1, 16, 11, 9, 48, 80, stride=(2, 2)),
    C.ConvCase("s2b", 3, 64, 15, 17, 64, 128, stride=(2, 2), dst_dt=C.S32),
    C.ConvCase("s21", 2, 32, 12, 9, 32, 32, stride=(2, 1), pad=(0, 1), dst_dt=C.S8, relu1=False),
    C.ConvCase("k5", 1, 16, 9, 9, 16, 16, k=(5, 5), pad=(2, 2)),
    C.ConvCase("k7s2", 2, 16, 23, 21, 64, 64, k=(7, 7), stride=(2, 2), pad=(3, 3), dst_dt=C.S32),
    C.ConvCase("k1", 2, 32, 5, 5, 32, 0, k=(1, 1), pad=(0, 0), dst_dt=C.S32),
    C.ConvCase("k1f", 4, 48, 6, 5, 80, 48, k=(1, 1), pad=(0, 0), dst_dt=C.F32, relu1=False),
    C.ConvCase("k13", 1, 32, 8, 10, 32, 64, k=(1, 3), pad=(0, 1), dst_dt=C.S32),
    C.ConvCase("k31", 1, 32, 8, 10, 48, 0, k=(3, 1), pad=(1, 0), dst_dt=C.S8, relu0=False),
    C.ConvCase("ic128", 1, 128, 6, 6, 80, 48, dst_dt=C.S32, wide=True),
    C.ConvCase("ic80", 2, 80, 7, 7, 96, 160, dst_dt=C.U8, wide=True),
    C.ConvCase("res3", 2, 128, 14, 14, 128, 512, dst_dt=C.S32),
    C.ConvCase("res3w", 1, 128, 9, 20, 128, 256, dst_dt=C.U8, wide=True, per_channel0=True, per_channel1=True),
    C.ConvCase("ic256", 1, 256, 7, 7, 256, 0, dst_dt=C.S32, wide=True, relu0=False),
    C.ConvCase("oc320", 5, 64, 7, 7, 320, 112, dst_dt=C.F32, rm0=1),
    C.ConvCase("img3", 7, 32, 3, 3, 32, 32, dst_dt=C.S32),
    C.ConvCase("one", 1, 16, 3, 3, 16, 16, pad=(0, 0)),
    C.ConvCase("w200", 1, 32, 3, 200, 32, 32, dst_dt=C.S32),
    C.ConvCase("w131", 2, 16, 5, 131, 16, 0, dst_dt=C.U8, pad=(1, 0)),
]
STREAM_CASES = (STREAM_SHAPES + C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL) + C.option_sweep(C.SMALL64)
                + [C.unfused(c) for c in C.dtype_matrix(C.SMALL) + C.option_sweep(C.SMALL64)]
                + [C.CONFIG2, C.CONFIG3_SMALL, C.unfused(C.REF_SHAPES[2])])


@pytest.mark.parametrize("case", STREAM_CASES, ids=lambda c: c.ident())
def test_stream_variant(hip, oracle, tuning, case):
    """the general-shape variant as the host picks it; where that is conv_direct.cuh (fused, >= 64 channels on both
    sides) also conv_stream.cuh's fused path, which stays the fallback for shapes no direct instance fits"""
    data = C.generate(case)
    ref = hip.oracle_conv(oracle, case, data)
    got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
    assert info.variant == hip.dfa.VARIANT_MFMA_STREAM, info.kernel_name
    hip.assert_bit_equal(got, ref, info.kernel_name.decode())
    if info.kernel_name.decode().startswith("conv_direct_kernel"):
        tuning.setenv("DFX_STREAM_DIRECT", "0")
        got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
        assert info.kernel_name.decode().startswith("conv_stream_kernel"), info.kernel_name
        hip.assert_bit_equal(got, ref, info.kernel_name.decode())


def test_stream_variant_exact_requant_path(hip, oracle, tuning):
    """DFX_NO_FAST=1 (read at set_weights) forces the exact requant code path on inputs the
    fast-path proof would otherwise accept."""
    tuning.setenv("DFX_STREAM_DIRECT", "0")   # conv_stream.cuh is the subject here (conv_direct.cuh: tests below)
    tuning.setenv("DFX_NO_FAST", "1")
    for case in STREAM_SHAPES[:12] + C.dtype_matrix(C.SMALL) + [C.unfused(c) for c in C.dtype_matrix(C.SMALL)]:
        data = C.generate(case)
        got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), case.ident())


@pytest.mark.parametrize("grid", ["1", "3"])
def test_stream_variant_many_units_per_workgroup(hip, oracle, tuning, grid):
    """DFX_STREAM_GRID caps the grid so that every workgroup walks many units: covers the
    cross-unit tile / weight prefetch and the unit-to-unit LDS reuse."""
    tuning.setenv("DFX_STREAM_DIRECT", "0")   # conv_stream.cuh is the subject here (conv_direct.cuh: tests below)
    tuning.setenv("DFX_STREAM_GRID", grid)
    for case in STREAM_SHAPES + [C.CONFIG3_SMALL, C.unfused(C.CONFIG2), replace(C.CONFIG3_SMALL, dst_dt=C.U8, wide=True)]:
        data = C.generate(case)
        got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
        assert info.grid <= int(grid)
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), case.ident())


STREAM_FULL = [
    C.ConvCase("res3", 128, 128, 28, 28, 128, 512, dst_dt=C.U8),
    C.ConvCase("res3w", 32, 128, 28, 28, 128, 512, dst_dt=C.S32, wide=True, per_channel0=True, per_channel1=True),
    C.ConvCase("res4", 64, 256, 14, 14, 256, 1024, dst_dt=C.U8),
    C.ConvCase("res5", 32, 512, 7, 7, 512, 2048, dst_dt=C.S8, relu1=False),
    C.ConvCase("res3s2", 32, 128, 56, 56, 128, 512, stride=(2, 2), dst_dt=C.F32),
    C.unfused(C.ConvCase("vgg128", 8, 128, 56, 56, 128, 0, dst_dt=C.U8)),
]


@pytest.mark.parametrize("case", STREAM_FULL, ids=lambda c: c.ident())
def test_stream_variant_full_size(hip, oracle, case):
    """ResNet-50 res3..res5-style blocks at (near) bench size: more units than workgroups (fused: conv_direct.cuh
    with the host's choice of waves, unit size and 1x1 split; unfused: conv_stream.cuh)."""
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant == hip.dfa.VARIANT_MFMA_STREAM, info.kernel_name
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode())


@pytest.mark.parametrize("planes", ["0", "1"])
def test_stream_variant_resident_input_chunks(hip, oracle, tuning, planes):
    """DFX_STREAM_PLANES: all 64-channel input chunks resident in LDS (staged once per work
    item) vs one chunk at a time, on inputs with several chunks."""
    tuning.setenv("DFX_STREAM_DIRECT", "0")   # conv_stream.cuh is the subject here (conv_direct.cuh: tests below)
    tuning.setenv("DFX_STREAM_PLANES", planes)
    cases = [c for c in STREAM_SHAPES if c.ic > 64] + [
        C.ConvCase("ic192", 3, 192, 9, 12, 160, 144, dst_dt=C.U8, wide=True),
        C.ConvCase("ic320", 2, 320, 5, 6, 272, 0, dst_dt=C.S32, k=(1, 1), pad=(0, 0)),
        C.ConvCase("ic144s2", 2, 144, 11, 9, 64, 256, stride=(2, 2), dst_dt=C.S8, relu1=False)]
    for grid in ("", "2"):
        if grid:
            tuning.setenv("DFX_STREAM_GRID", grid)
        for case in cases:
            data = C.generate(case)
            got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
            hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode() + " " + case.ident())


def test_stream_variant_chunk_parallel(hip, oracle, tuning):
    """(unit, output chunk) work items for unfused ops, forced on shapes that would not pick them.  (The two-launch
    execution of fused ops that round 2 tested here is gone: conv_direct.cuh serves those shapes in one launch.)"""
    tuning.setenv("DFX_STREAM_DIRECT", "0")   # conv_stream.cuh is the subject here (conv_direct.cuh: tests below)
    tuning.setenv("DFX_STREAM_OCC_PAR", "1")
    cases = STREAM_SHAPES + C.dtype_matrix(C.SMALL) + [C.CONFIG3_SMALL, replace(C.CONFIG3_SMALL, dst_dt=C.U8, wide=True)]
    cases = [C.unfused(c) if c.oc1x1 else c for c in cases] + [
        C.ConvCase("oc512", 3, 64, 7, 7, 512, 0, dst_dt=C.S8, relu0=False),
        C.ConvCase("oc320s2", 2, 48, 9, 11, 320, 0, stride=(2, 2), dst_dt=C.F32)]
    for grid in ("", "2"):
        if grid:
            tuning.setenv("DFX_STREAM_GRID", grid)
        for case in cases:
            data = C.generate(case)
            got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
            hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode() + " " + case.ident())


DIRECT_SHAPES = [
    C.ConvCase("res3", 2, 128, 14, 14, 128, 512, dst_dt=C.S32),
    C.ConvCase("res3u", 3, 128, 9, 20, 128, 256, dst_dt=C.U8, wide=True, per_channel0=True, per_channel1=True),
    C.ConvCase("d64", 2, 64, 12, 10, 64, 64, dst_dt=C.S8, relu1=False),
    C.ConvCase("d96", 2, 48, 7, 9, 96, 160, dst_dt=C.F32, rm0=1, rm1=1),
    C.ConvCase("d192s2", 2, 80, 13, 11, 192, 320, stride=(2, 2), dst_dt=C.U8),
    C.ConvCase("d256k1", 5, 96, 6, 5, 256, 128, k=(1, 1), pad=(0, 0), dst_dt=C.S32),
    C.ConvCase("d128k5", 1, 16, 9, 9, 128, 96, k=(5, 5), pad=(2, 2), dst_dt=C.U8),
    C.ConvCase("d7x7", 9, 64, 7, 7, 64, 256, dst_dt=C.S32),
    C.ConvCase("dw200", 1, 32, 3, 200, 64, 64, dst_dt=C.S32),
    C.ConvCase("d512", 2, 256, 7, 7, 512, 640, dst_dt=C.U8, wide=True),
]


@pytest.mark.parametrize("grid", ["", "2"])
def test_direct_weight_kernel(hip, oracle, tuning, grid):
    """fused ops with >= 64 channels on both sides run on conv_direct.cuh (weights straight from
    L2 into MFMA operands); also with DFX_NO_FAST (exact requant path) and many units per workgroup."""
    if grid:
        tuning.setenv("DFX_STREAM_GRID", grid)
    for nofast in ("0", "1"):
        tuning.setenv("DFX_NO_FAST", nofast)
        for case in DIRECT_SHAPES + [C.CONFIG3_SMALL, replace(C.CONFIG3_SMALL, dst_dt=C.U8, wide=True)]:
            data = C.generate(case)
            got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
            assert info.kernel_name.decode().startswith("conv_direct_kernel"), info.kernel_name
            hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode() + " " + case.ident())


@pytest.mark.parametrize("split", ["nw4 wo1=1 npb4", "nw4 wo1=4 npb4", "nw4 wo1=4 npb2", "nw4 wo1=4 npb1",
                                   "nw8 wo1=8 npb4", "nw8 wo1=8 npb2", "nw8 wo1=8 npb1"])
@pytest.mark.parametrize("switch", [None, "DFX_NO_MAGIC", "DFX_NO_FAST"])
def test_direct_weight_kernel_unit_sizes(hip, oracle, tuning, split, switch):
    """conv_direct.cuh with four or eight waves per workgroup (DFX_DIRECT_NW), the 1x1 stage split by pixel blocks
    or by channel groups (DFX_DIRECT_WO1) and 1, 2 or 4 pixel blocks per unit (DFX_DIRECT_NPB; the host picks all
    three by the number of units), and its three requant routes: fma / magic without conversions (default where
    the host can prove the ranges), fast (DFX_NO_MAGIC), exact (DFX_NO_FAST)."""
    nw, wo1, npb = split[2], split.split("wo1=")[1][0], split[-1]
    tuning.setenv("DFX_STREAM_DIRECT", "1")
    tuning.setenv("DFX_DIRECT_NW", nw)
    tuning.setenv("DFX_DIRECT_WO1", wo1)
    tuning.setenv("DFX_DIRECT_NPB", npb)
    if switch:
        tuning.setenv(switch, "1")
    cases = DIRECT_SHAPES + [replace(C.CONFIG3_SMALL, dst_dt=C.U8), replace(C.CONFIG3_SMALL, dst_dt=C.U8, wide=True),
                             C.ConvCase("d14", 3, 256, 14, 14, 256, 512, dst_dt=C.U8),
                             C.ConvCase("d7", 5, 128, 7, 7, 192, 384, dst_dt=C.U8, per_channel0=True, per_channel1=True),
                             C.ConvCase("d7s8", 2, 128, 7, 7, 128, 256, dst_dt=C.S8, relu1=False),
                             C.ConvCase("d28s32", 1, 64, 28, 28, 128, 128, dst_dt=C.S32)]
    for case in cases:
        data = C.generate(case)
        got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
        name = info.kernel_name.decode()
        if not name.startswith("conv_direct_kernel"):
            # no instance of this split for the shape (eight waves need a multiple of 8 output blocks and groups
            # of four; wo1 = 4 needs a multiple of 4 groups): the op falls back to conv_stream.cuh
            assert nw == "8" or wo1 == "4", name
        else:
            assert name.startswith("conv_direct_kernel<nw%s," % nw) and name.endswith("npb%s>" % npb), name
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), name + " " + case.ident())


UNFUSED_DIRECT_SHAPES = [
    C.ConvCase("u64", 3, 64, 12, 10, 64, 0, dst_dt=C.U8),
    C.ConvCase("u64s8", 2, 64, 9, 20, 64, 0, dst_dt=C.S8, relu0=False, wide=True),
    C.ConvCase("u128s32", 2, 128, 14, 14, 256, 0, dst_dt=C.S32, per_channel0=True),
    C.ConvCase("u96f32", 2, 64, 7, 9, 96, 0, dst_dt=C.F32, rm0=1),                      # exact route, 96 = 3 output blocks (padded to WO)
    C.ConvCase("u256", 3, 256, 14, 14, 256, 0, dst_dt=C.U8, per_channel0=True),        # eight output blocks: the 8-wave forms
    C.ConvCase("us2", 2, 64, 23, 21, 128, 0, stride=(2, 2), dst_dt=C.U8, wide=True),
    C.ConvCase("u5x5", 2, 64, 12, 10, 64, 0, k=(5, 5), pad=(2, 2), dst_dt=C.S8),
    C.ConvCase("unopad", 5, 64, 8, 8, 128, 0, pad=(0, 0), dst_dt=C.U8, bia0_dt=C.UNDEF),
    # deep pointwise convs whose weights do not fit conv_pw.cuh's LDS image: 16 input planes, the most the staging table holds
    C.ConvCase("upw1024", 3, 1024, 5, 5, 256, 0, k=(1, 1), pad=(0, 0), dst_dt=C.U8),
    C.ConvCase("upw512s32", 2, 512, 6, 7, 256, 0, k=(1, 1), pad=(0, 0), dst_dt=C.S32, relu0=False),
    # u8 dst without ReLU on the exact route (round-down): negative results saturate at 0 (found by profiles/debug/soak_random.py)
    C.ConvCase("u8norelu", 21, 192, 4, 3, 512, 0, k=(3, 2), stride=(1, 2), pad=(1, 1), dst_dt=C.U8, bia0_dt=C.S8, relu0=False, rm0=1),
]


@pytest.mark.parametrize("split", ["nw4 wo1=1 npb4", "nw4 wo1=4 npb2", "nw4 wo1=4 npb1", "nw8 wo1=8 npb4", "nw8 wo1=8 npb2", "nw8 wo1=8 npb1"])
@pytest.mark.parametrize("switch", [None, "DFX_NO_MAGIC", "DFX_NO_FAST"])
def test_direct_weight_kernel_unfused(hip, oracle, tuning, split, switch):
    """unfused convs with a window and >= 64 channels on both sides run on conv_direct.cuh without its 1x1 stage
    (DirectGeom::unfused): 4-byte outputs stored straight from the accumulators, 1-byte outputs collected in LDS in
    natural channel order and written as whole pixel rows; all dst types, the three requant routes (stage-0 "fma"
    for u8, fast, exact), every wave / unit split."""
    nw, wo1, npb = split[2], split.split("wo1=")[1][0], split[-1]
    tuning.setenv("DFX_DIRECT_NW", nw)
    tuning.setenv("DFX_DIRECT_WO1", wo1)
    tuning.setenv("DFX_DIRECT_NPB", npb)
    if switch:
        tuning.setenv(switch, "1")
    ran = 0
    for case in UNFUSED_DIRECT_SHAPES:
        data = C.generate(case)
        got, info = hip.hip_conv(case, data, force_variant=hip.dfa.VARIANT_MFMA_STREAM)
        name = info.kernel_name.decode()
        if name.startswith("conv_direct_kernel"):
            assert "unfused," in name and name.startswith("conv_direct_kernel<nw%s," % nw) and name.endswith("npb%s>" % npb), name
            ran += 1
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), name + " " + case.ident())
    assert ran >= (1 if nw == "8" else 4), ran


PW_SHAPES = [
    C.ConvCase("pw256_64", 3, 256, 9, 11, 64, 0, k=(1, 1), pad=(0, 0), dst_dt=C.U8),                    # 297 px: partial last block
    C.ConvCase("pw512_128", 2, 512, 7, 7, 128, 0, k=(1, 1), pad=(0, 0), dst_dt=C.S8, relu0=False, wide=True),
    C.ConvCase("pw256_256", 2, 256, 5, 13, 256, 0, k=(1, 1), pad=(0, 0), dst_dt=C.S32, per_channel0=True),
    C.ConvCase("pw768_64", 1, 768, 6, 6, 64, 0, k=(1, 1), pad=(0, 0), dst_dt=C.F32, rm0=1),
    C.ConvCase("pw256_128", 5, 256, 4, 4, 128, 0, k=(1, 1), pad=(0, 0), dst_dt=C.U8, relu0=False, rm0=1, bia0_dt=C.S8),
    C.ConvCase("pw1px", 1, 256, 1, 1, 64, 0, k=(1, 1), pad=(0, 0), dst_dt=C.U8),                          # one pixel
    C.ConvCase("pw_many", 2, 256, 40, 37, 64, 0, k=(1, 1), pad=(0, 0), dst_dt=C.U8, bia0_dt=C.UNDEF),    # more blocks than waves
]


@pytest.mark.parametrize("switch", [None, "DFX_NO_MAGIC", "DFX_NO_FAST"])
def test_pointwise_kernel(hip, oracle, tuning, switch):
    """unfused 1x1 stride-1 convs whose weights fit LDS (ic a multiple of 256, oc 64 / 128 / 256) run on conv_pw.cuh:
    pixel fragments straight from global memory into the MFMA operands; all dst types, the three requant routes;
    DFX_STREAM_PW=0 sends the same shapes to conv_stream.cuh."""
    if switch:
        tuning.setenv(switch, "1")
    for case in PW_SHAPES:
        data = C.generate(case)
        ref = hip.oracle_conv(oracle, case, data)
        got, info = hip.hip_conv(case, data)
        assert info.kernel_name.decode().startswith("conv_pw_kernel"), info.kernel_name
        hip.assert_bit_equal(got, ref, info.kernel_name.decode() + " " + case.ident())
    tuning.setenv("DFX_STREAM_PW", "0")
    case = PW_SHAPES[0]
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.kernel_name.decode().startswith("conv_stream_kernel"), info.kernel_name
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), info.kernel_name.decode())


def test_dst_of_4_gib_is_named_in_kernel_name(hip):
    """the streamed- / direct-weight MFMA kernels keep dst offsets in 32 bits: an op whose dst reaches 4 GiB is
    created on the scalar kernel and dfx_conv_query says so (create only: nothing of that size is allocated)"""
    # 66 x 128 x 128 px x 1024 channels x 4 B = 4.1 GiB
    op = hip.dfa.Conv((66, 128, 128, 128), (128, 128, 3, 3), dst_dt=C.F32, oc1x1=1024)
    try:
        name = op.info().kernel_name.decode()
        assert name.startswith("conv_generic_kernel") and "dst >= 4 GiB" in name, name
    finally:
        op.close()
    op = hip.dfa.Conv((8, 128, 128, 128), (128, 128, 3, 3), dst_dt=C.F32, oc1x1=1024)   # the same layer under the limit
    try:
        assert op.info().kernel_name.decode().startswith("conv_direct_kernel"), op.info().kernel_name
    finally:
        op.close()


def _random_cases(n, seed, big=False):
    """seeded random shapes inside what the reference's init_conf admits: channels multiples
    of 16, any kernel / stride / padding with a non-empty output"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        sh, sw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        ph, pw = int(rng.integers(0, kh)), int(rng.integers(0, kw))
        ih, iw = (int(rng.integers(1, 13)), int(rng.integers(1, 13))) if big else \
                 (int(rng.integers(1, 24)), int(rng.integers(1, 40)))
        if ih + 2 * ph < kh or iw + 2 * pw < kw:
            continue
        fused = bool(rng.integers(0, 4))
        cmax = (21, 33, 41) if big else (9, 11, 13)   # channel counts / 16 (exclusive)
        out.append(C.ConvCase(
            "rnd%d" % len(out), int(rng.integers(1, 41 if big else 6)), 16 * int(rng.integers(1, cmax[0])), ih, iw,
            16 * int(rng.integers(1, cmax[1])), 16 * int(rng.integers(1, cmax[2])) if fused else 0,
            k=(kh, kw), stride=(sh, sw), pad=(ph, pw),
            dst_dt=int(rng.choice([C.U8, C.S8, C.S32, C.F32])),
            bia0_dt=int(rng.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
            bia1_dt=int(rng.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
            relu0=bool(rng.integers(0, 2)), relu1=bool(rng.integers(0, 2)),
            rm0=int(rng.integers(0, 2)), rm1=int(rng.integers(0, 2)),
            per_channel0=bool(rng.integers(0, 2)), per_channel1=bool(rng.integers(0, 2)),
            wide=bool(rng.integers(0, 2)), seed=1000 + len(out)))
    return out


def test_random_shapes_all_variants(hip, oracle):
    """90 seeded random shapes / option sets (30 of them many-channel, many-image): the streamed-weight MFMA kernel (auto) and the
    scalar kernel both reproduce the oracle bit for bit."""
    for case in _random_cases(60, 2024) + _random_cases(30, 7, big=True):
        data = C.generate(case)
        ref = hip.oracle_conv(oracle, case, data)
        for fv in (-1, hip.dfa.VARIANT_GENERIC):
            got, info = hip.hip_conv(case, data, force_variant=fv)
            hip.assert_bit_equal(got, ref, "%s %r" % (info.kernel_name.decode(), case))


def test_auto_variant_prefers_mfma(hip):
    """shapes outside the resident-weight kernel go to the streamed MFMA kernel, not the scalar one."""
    for case in STREAM_SHAPES[:4]:
        op = hip.make_conv(case, C.generate(case))
        assert op.info().variant == hip.dfa.VARIANT_MFMA_STREAM
        op.close()


def test_host_buffer_path(hip, oracle):
    """drop-in semantics of op::submit(): host pointers in, host pointers out."""
    for case in (C.CONFIG2, C.unfused(C.SMALL)):
        data = C.generate(case)
        got, info = hip.hip_conv(case, data, host_path=True)
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "host path")


def test_golden_vectors(hip):
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "conv_*.npz")))
    assert files
    for f in files:
        z = np.load(f, allow_pickle=False)
        case = C.ConvCase(**{k: (tuple(v.tolist()) if v.ndim else v.item())
                             for k, v in ((k[5:], z[k]) for k in z.files if k.startswith("case_"))})
        data = dict(src=z["src"], w0=z["w0"], w1=z["w1"] if "w1" in z.files else None,
                    bia0=z["bia0"] if "bia0" in z.files else None,
                    bia1=z["bia1"] if "bia1" in z.files else None,
                    scales0=z["scales0"], scales1=z["scales1"])
        for fv in (-1, hip.dfa.VARIANT_GENERIC):
            got, info = hip.hip_conv(case, data, force_variant=fv)
            hip.assert_bit_equal(got, z["dst"], os.path.basename(f) + " " + info.kernel_name.decode())


def test_cvt_edge_cases(hip, oracle):
    """NaN/overflow -> 0x80000000 (vcvtps2dq), -0.0 through vmaxps, on both variants."""
    case = C.ConvCase("edge", 1, 32, 5, 5, 32, 32, dst_dt=C.S32, bia0_dt=C.UNDEF,
                      bia1_dt=C.UNDEF, relu1=False, wide=True)
    data = C.generate(case)
    data["scales1"] = np.array([3.0e38], dtype=np.float32)
    ref = hip.oracle_conv(oracle, case, data)
    assert (ref == -2147483648).any()
    for fv in (-1, hip.dfa.VARIANT_GENERIC):
        hip.assert_bit_equal(hip.hip_conv(case, data, force_variant=fv)[0], ref, "overflow")
    for dst_dt in (C.U8, C.S8):
        c2 = replace(case, dst_dt=dst_dt)
        ref = hip.oracle_conv(oracle, c2, data)
        for fv in (-1, hip.dfa.VARIANT_GENERIC):
            hip.assert_bit_equal(hip.hip_conv(c2, data, force_variant=fv)[0], ref, "overflow narrow")
    case = C.ConvCase("negzero", 1, 32, 5, 5, 32, 32, dst_dt=C.F32, bia0_dt=C.UNDEF,
                      bia1_dt=C.UNDEF, relu1=True)
    data = C.generate(case)
    data["scales1"] = np.array([-0.0], dtype=np.float32)
    ref = hip.oracle_conv(oracle, case, data)
    assert (ref.view(np.uint32) == 0x80000000).any()
    for fv in (-1, hip.dfa.VARIANT_GENERIC):
        hip.assert_bit_equal(hip.hip_conv(case, data, force_variant=fv)[0], ref, "-0.0")
    data["scales0"] = np.array([np.nan], dtype=np.float32)       # NaN intermediate -> 255
    ref = hip.oracle_conv(oracle, case, data)
    for fv in (-1, hip.dfa.VARIANT_GENERIC):
        hip.assert_bit_equal(hip.hip_conv(case, data, force_variant=fv)[0], ref, "nan")


def test_full_size_config3(hip, oracle):
    """BASELINE.json configs[2] at full size (N=128, 56x56, 64->64->256, s32 out):
    complete comparison with the multi-threaded oracle plus the size-independent
    property that images are independent (batch-sharding invariant)."""
    case = replace(C.CONFIG3_SMALL, name="cfg3", bs=128)
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant == hip.dfa.VARIANT_MFMA_FUSED
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "config 3 full")
    # shard invariance: images [40, 56) alone give the same bytes
    sub = dict(data, src=data["src"][40:56])
    got_sub, _ = hip.hip_conv(replace(case, bs=16), sub)
    hip.assert_bit_equal(got_sub, got[40:56], "batch shard")


def test_full_size_config5_rows(hip, oracle):
    """configs[4] shape (224x224, 64->64->128, f32 out) at N=2: wide-row units."""
    case = replace(C.CONFIG5_TINY, bs=2)
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.variant == hip.dfa.VARIANT_MFMA_FUSED
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "config 5")


def test_full_size_config5(hip, oracle):
    """BASELINE.json configs[4] at FULL size (N=64, 224x224, 64->64->128, f32 out; 1.64 GB of dst, 24 units per
    loader through the lazy device queue): the HIP path at full size, the oracle on images {0,1,31,32,62,63}
    (images are independent, op_conv.cc:155-156), and the size-independent batch-shard property on images
    [20, 24).  f32 compared bit for bit (0 ulp; north_star allows 1)."""
    import torch
    case = replace(C.CONFIG5_TINY, name="cfg5", bs=64)
    rng = np.random.default_rng(case.seed)
    data = C.generate(replace(case, bs=1))            # weights, bias, scales (seeded; independent of bs)
    data["src"] = rng.integers(0, 17, (case.bs, case.ih, case.iw, case.ic), dtype=np.uint8)
    op = hip.make_conv(case, data)
    info = op.info()
    assert info.variant == hip.dfa.VARIANT_MFMA_FUSED
    src = torch.from_numpy(data["src"]).cuda()
    dst = torch.empty(op.dst_shape, dtype=torch.float32, device="cuda")
    dst.view(torch.uint8).fill_(0xCD)
    op.submit(src, dst)
    torch.cuda.synchronize()
    op.close()
    pick = [0, 1, 31, 32, 62, 63]
    ref = hip.oracle_conv(oracle, replace(case, bs=len(pick)), dict(data, src=np.ascontiguousarray(data["src"][pick])))
    got = dst[pick].cpu().numpy()
    hip.assert_bit_equal(got, ref, "config 5 full size, images %s" % pick)
    # no element of the 1.64 GB left unwritten (poison pattern 0xCDCDCDCD is a float no requant produces here)
    assert int((dst.view(torch.int32) == -842150451).sum().item()) == 0
    sub_case = replace(case, bs=4)
    got_sub, _ = hip.hip_conv(sub_case, dict(data, src=np.ascontiguousarray(data["src"][20:24])))
    hip.assert_bit_equal(got_sub, dst[20:24].cpu().numpy(), "batch shard [20,24)")


def test_config4_shaped_gather_and_concat(hip, oracle):
    """BASELINE.json configs[3] on ONE GPU: the eight ranks' 16-image s32 outputs of the res2a block (each
    checked against the oracle) laid out rank-major exactly as the all-gather leaves them
    (dist.gathered_offsets) -> dfx_concat_submit_gathered at {16,56,56,8x256} + ReLU, against oracle.concat."""
    import importlib
    import torch
    dist = importlib.import_module("deep-fusion_amd.dist")
    world, per_rank = 8, 16
    case = replace(C.CONFIG3_SMALL, name="cfg4r", bs=per_rank, relu1=False)   # negative s32 values reach the ReLU
    data = C.generate(case)
    rng = np.random.default_rng(77)
    op = hip.make_conv(case, data)
    chans = [case.oc1x1] * world
    offs, total = dist.gathered_offsets(per_rank, case.oh, case.ow, chans, 4)
    gathered = torch.empty(total, dtype=torch.uint8, device="cuda")
    shards = []
    for r in range(world):
        src_np = rng.integers(0, 17, data["src"].shape, dtype=np.uint8)
        view = gathered[offs[r]:offs[r] + per_rank * case.oh * case.ow * case.oc1x1 * 4].view(torch.int32)
        op.submit(torch.from_numpy(src_np).cuda(), view.view(per_rank, case.oh, case.ow, case.oc1x1))
        torch.cuda.synchronize()
        shard = view.view(per_rank, case.oh, case.ow, case.oc1x1).cpu().numpy()
        if r in (0, 7):
            hip.assert_bit_equal(shard, hip.oracle_conv(oracle, case, dict(data, src=src_np)), "rank %d conv" % r)
        shards.append(shard)
    op.close()
    assert min(int(s.min()) for s in shards) < 0
    cat = hip.dfa.Concat(per_rank, case.oh, case.ow, chans, np.int32, True)
    dst = torch.empty(cat.dst_shape, dtype=torch.int32, device="cuda")
    dst.view(torch.uint8).fill_(0xCD)
    cat.submit_gathered(gathered, offs, dst)
    torch.cuda.synchronize()
    hip.assert_bit_equal(dst.cpu().numpy(), oracle.concat(shards, True), "configs[3] gathered concat")


def test_concat(hip, oracle):
    import torch
    from test_oracle import CONCAT_SHAPES, CONCAT_NARROW, concat_inputs
    for np_dt in (np.float32, np.int32, np.int8, np.uint8):
        shapes = CONCAT_SHAPES + (CONCAT_NARROW if np_dt in (np.float32, np.int32) else [])
        for sh in shapes:
            for wide in (False, True):
                srcs = concat_inputs(sh, np_dt, wide)
                n, _, h, w = sh[0]
                for relu in (True, False):
                    op = hip.dfa.Concat(n, h, w, [s[1] for s in sh], np_dt, relu)
                    ref = oracle.concat(srcs, relu)
                    dsrcs = [torch.from_numpy(s).cuda() for s in srcs]
                    dst = torch.empty(op.dst_shape, dtype=dsrcs[0].dtype, device="cuda")
                    dst.view(torch.uint8).fill_(0xCD)
                    op.submit(dsrcs, dst)
                    torch.cuda.synchronize()
                    hip.assert_bit_equal(dst.cpu().numpy(), ref, "concat dev")
                    hip.assert_bit_equal(op.submit_host(srcs), ref, "concat host")
                    op.close()


def test_maxpool_and_eltwise_sum(hip, oracle):
    """the reference's roadmap ops through the C ABI (dfx_pool_*, dfx_eltwise_*), bit-exact against the
    oracle (parity unpinned: the reference ships no implementation, test/test_conv_relu_pooling.cc:231-281)."""
    import torch
    from test_oracle import POOL_CASES, pool_input
    for np_dt in (np.uint8, np.int8, np.int32, np.float32):
        for shape, k, s, p, o in POOL_CASES:
            x = pool_input(shape, np_dt, seed=3)
            op = hip.dfa.Pool(shape[0], shape[3], shape[1], shape[2], o[0], o[1], k, s, p, np_dt)
            src = torch.from_numpy(x).cuda()
            dst = torch.empty(op.dst_shape, dtype=src.dtype, device="cuda")
            dst.view(torch.uint8).fill_(0xCD)
            op.submit(src, dst)
            torch.cuda.synchronize()
            hip.assert_bit_equal(dst.cpu().numpy(), oracle.maxpool(x, k, s, p, o), "maxpool %s %s" % (shape, np_dt))
            op.close()
            for algo, inc in ((hip.dfa.Pool.AVG_INCLUDE_PADDING, True), (hip.dfa.Pool.AVG_EXCLUDE_PADDING, False)):
                op = hip.dfa.Pool(shape[0], shape[3], shape[1], shape[2], o[0], o[1], k, s, p, np_dt, algo=algo)
                dst.view(torch.uint8).fill_(0xCD)
                op.submit(src, dst)
                torch.cuda.synchronize()
                hip.assert_bit_equal(dst.cpu().numpy(), oracle.avgpool(x, k, s, p, o, inc),
                                     "avgpool %s %s include_padding=%s" % (shape, np_dt, inc))
                op.close()
        for shape, n in (((2, 5, 7, 24), 3), ((1, 1, 1, 1), 2), ((3, 9, 4, 17), 8), ((4, 56, 56, 64), 2)):
            xs = [pool_input(shape, np_dt, seed=20 + i) for i in range(n)]
            for relu in (False, True):
                op = hip.dfa.EltwiseSum(n, xs[0].size, np_dt, relu)
                dsrcs = [torch.from_numpy(v).cuda() for v in xs]
                dst = torch.empty(shape, dtype=dsrcs[0].dtype, device="cuda")
                dst.view(torch.uint8).fill_(0xCD)
                op.submit(dsrcs, dst)
                torch.cuda.synchronize()
                hip.assert_bit_equal(dst.cpu().numpy(), oracle.eltwise_sum(xs, relu), "eltwise %s %s" % (shape, np_dt))
                op.close()
    with pytest.raises(hip.dfa.DfxError):  # an output window entirely in the padding
        hip.dfa.Pool(1, 16, 4, 4, 4, 4, (2, 2), (2, 2), (0, 0), np.uint8)


POOL_FUSED_CASES = [
    # unfused 3x3 stride-1 convs with even output sizes: widths that are / are not multiples of 16, one and
    # two 32-channel blocks, every dst type, both requant routes (reference-range and wide weights), round down
    C.ConvCase("pf_u8", 3, 64, 12, 40, 64, 0, dst_dt=C.U8),
    C.ConvCase("pf_u8w", 2, 32, 8, 56, 64, 0, dst_dt=C.U8, wide=True),
    C.ConvCase("pf_s8", 2, 64, 6, 18, 32, 0, dst_dt=C.S8, relu0=False),
    C.ConvCase("pf_s32", 2, 32, 10, 34, 32, 0, dst_dt=C.S32, relu0=False),
    C.ConvCase("pf_f32", 2, 64, 14, 14, 64, 0, dst_dt=C.F32, relu0=False, per_channel0=True),
    C.ConvCase("pf_f32r", 5, 32, 4, 100, 64, 0, dst_dt=C.F32),
    C.ConvCase("pf_rd", 2, 64, 8, 24, 64, 0, dst_dt=C.U8, rm0=1, wide=True),
    C.ConvCase("pf_np", 2, 32, 10, 10, 32, 0, dst_dt=C.U8, pad=(0, 0)),
    C.ConvCase("pf_big", 9, 64, 56, 56, 64, 0, dst_dt=C.U8),
    C.ConvCase("pf_wide", 2, 64, 8, 224, 64, 0, dst_dt=C.U8),          # rows too wide for LDS: column-split units
    C.ConvCase("pf_w200", 1, 32, 6, 200, 32, 0, dst_dt=C.S32, relu0=False),
    C.ConvCase("pf_w72", 2, 64, 4, 72, 32, 0, dst_dt=C.S8),
]


@pytest.mark.parametrize("case", POOL_FUSED_CASES, ids=lambda c: c.ident())
def test_conv_with_fused_max_pool(hip, oracle, tuning, case):
    """dfx_conv_desc::fuse_pool = 2: the conv kernel's store stage takes the 2x2 maximum (tiles of 2 rows x 16
    columns); must equal oracle conv -> oracle 2x2/2 max pooling bit for bit, in every requant mode."""
    import torch
    data = C.generate(case)
    mid = hip.oracle_conv(oracle, case, data)
    ref = oracle.maxpool(mid, (2, 2), (2, 2), (0, 0), (mid.shape[1] // 2, mid.shape[2] // 2))
    for switch in (None, "DFX_NO_MAGIC", "DFX_NO_FAST"):
        if switch:
            tuning.setenv(switch, "1")
        op = hip.make_conv(case, data, fuse_pool=2)
        tuning.undo()
        assert op.dst_shape == ref.shape
        src = torch.from_numpy(data["src"]).cuda()
        tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
        dst = torch.empty(op.dst_shape, dtype=tdt, device="cuda")
        dst.view(torch.uint8).fill_(0xCD)
        op.submit(src, dst)
        torch.cuda.synchronize()
        hip.assert_bit_equal(dst.cpu().numpy(), ref, "fused pool %s %s" % (case.name, switch))
        op.close()


def test_fused_pool_is_refused_where_unsupported(hip):
    with pytest.raises(hip.dfa.DfxError):   # odd conv output size
        hip.dfa.Conv((1, 11, 9, 32), (32, 32, 3, 3), fuse_pool=2)
    with pytest.raises(hip.dfa.DfxError):   # fused 1x1 stage
        hip.dfa.Conv((1, 8, 8, 32), (32, 32, 3, 3), oc1x1=64, fuse_pool=2)
    with pytest.raises(hip.dfa.DfxError):   # not on the resident-weight kernel
        hip.dfa.Conv((1, 8, 8, 128), (128, 128, 3, 3), fuse_pool=2)


def test_concat_golden(hip):
    import torch
    for np_dt in (np.float32, np.int32, np.int8, np.uint8):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden",
                                 "concat_%s.npz" % np.dtype(np_dt).name), allow_pickle=False)
        si = 0
        while "s%d_src0" % si in z.files:
            srcs = []
            while "s%d_src%d" % (si, len(srcs)) in z.files:
                srcs.append(z["s%d_src%d" % (si, len(srcs))])
            n, h, w, _ = srcs[0].shape
            for relu in (0, 1):
                op = hip.dfa.Concat(n, h, w, [s.shape[3] for s in srcs], np_dt, bool(relu))
                hip.assert_bit_equal(op.submit_host(srcs), z["s%d_relu%d_dst" % (si, relu)], "concat golden")
                op.close()
            si += 1


def test_concat_gathered(hip, oracle):
    """rank-major staging buffer as an all-gather leaves it."""
    import torch
    rng = np.random.default_rng(5)
    shards = [rng.integers(0, 256, (3, 5, 7, c)).astype(np.uint8) for c in (32, 64, 16, 48)]
    flat = np.concatenate([s.reshape(-1) for s in shards])
    offs = np.cumsum([0] + [s.size for s in shards[:-1]]).tolist()
    op = hip.dfa.Concat(3, 5, 7, [s.shape[3] for s in shards], np.uint8, True)
    g = torch.from_numpy(flat).cuda()
    dst = torch.empty(op.dst_shape, dtype=torch.uint8, device="cuda")
    op.submit_gathered(g, offs, dst)
    torch.cuda.synchronize()
    hip.assert_bit_equal(dst.cpu().numpy(), oracle.concat(shards, True), "gathered")


def test_repeated_submits_rearm_queue(hip, oracle):
    """The persistent MFMA kernel pulls units from a device-side queue that the last
    workgroup re-arms; back-to-back submits of one op must all be complete and equal."""
    import torch
    for case in (replace(C.CONFIG3_SMALL, bs=5), C.CONFIG2, C.ConvCase("w96", 3, 64, 6, 96, 64, 128, dst_dt=C.S32)):
        data = C.generate(case)
        ref = hip.oracle_conv(oracle, case, data)
        op = hip.make_conv(case, data)
        src = torch.from_numpy(data["src"]).cuda()
        tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
        outs = [torch.empty(op.dst_shape, dtype=tdt, device="cuda") for _ in range(6)]
        for o in outs:
            o.view(torch.uint8).fill_(0xCD)
        for o in outs:
            op.submit(src, o)
        torch.cuda.synchronize()
        for o in outs:
            hip.assert_bit_equal(o.cpu().numpy(), ref, "repeat " + case.name)
        op.close()


def test_concurrent_submits_on_several_streams(hip, oracle, tuning):
    """one handle, launches in flight on many streams at once: every launch has its own argument block and its
    own slot of the 16-entry queue ring (include/dfx.h), so the outputs must not mix.  20 streams x 3 launches =
    60 launches wrap the ring several times: a launch whose slot was last used on another stream waits for that
    launch on the device (slot events) instead of sharing its queue words.  Different inputs per stream; the
    resident-weight kernel (device queue: many units per loader), a streamed one, and a direct-weight one
    (eight waves) submitted first on the default stream and then on side streams."""
    import torch
    cases = [(C.ConvCase("conc_q", 6, 32, 120, 96, 64, 0, dst_dt=C.F32), 20, None),
             (replace(C.CONFIG3_SMALL, bs=5, dst_dt=C.U8), 20, None),
             (C.ConvCase("conc_s", 3, 128, 14, 14, 128, 256, dst_dt=C.S32), 4, None),
             (C.ConvCase("conc_direct", 2, 128, 7, 7, 256, 512, dst_dt=C.U8), 4, "DFX_DIRECT_NW=8")]
    for case, nstreams, switch in cases:
        if switch:
            tuning.setenv(*switch.split("="))
        data = C.generate(case)
        op = hip.make_conv(case, data)
        if switch:
            assert op.info().kernel_name.decode().startswith("conv_direct_kernel<nw8"), op.info().kernel_name
        rng = np.random.default_rng(11)
        nsrc = min(nstreams, 5)   # distinct inputs (stream i uses input i % nsrc)
        srcs_np = [rng.integers(0, 256, data["src"].shape).astype(np.uint8) for _ in range(nsrc)]
        refs = [hip.oracle_conv(oracle, case, dict(data, src=sn)) for sn in srcs_np]
        streams = [None if (switch and i == 0) else torch.cuda.Stream() for i in range(nstreams)]
        tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
        srcs = [torch.from_numpy(sn).cuda() for sn in srcs_np]
        outs = [[torch.empty(op.dst_shape, dtype=tdt, device="cuda") for _ in range(3)] for _ in range(nstreams)]
        for row in outs:
            for o in row:
                o.view(torch.uint8).fill_(0xCD)
        torch.cuda.synchronize()
        for rep in range(3):
            for i, st in enumerate(streams):
                op.submit(srcs[i % nsrc], outs[i][rep], stream=st)
        torch.cuda.synchronize()
        for i in range(nstreams):
            for rep in range(3):
                hip.assert_bit_equal(outs[i][rep].cpu().numpy(), refs[i % nsrc],
                                     "stream %d launch %d %s" % (i, rep, case.name))
        op.close()
        if switch:
            tuning.setenv(switch.split("=")[0], None)


def test_concurrent_submits_from_two_host_threads(hip, oracle):
    """include/dfx.h: a handle may be submitted from several host threads.  Two threads, each with its own
    stream and input, 24 launches each (the ring wraps three times)."""
    import threading
    import torch
    case = replace(C.CONFIG3_SMALL, bs=5, dst_dt=C.U8)
    data = C.generate(case)
    op = hip.make_conv(case, data)
    rng = np.random.default_rng(12)
    srcs_np = [rng.integers(0, 256, data["src"].shape).astype(np.uint8) for _ in range(2)]
    refs = [hip.oracle_conv(oracle, case, dict(data, src=sn)) for sn in srcs_np]
    srcs = [torch.from_numpy(sn).cuda() for sn in srcs_np]
    outs = [[torch.empty(op.dst_shape, dtype=torch.uint8, device="cuda") for _ in range(24)] for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    errs = []

    def work(i):
        try:
            for o in outs[i]:
                op.submit(srcs[i], o, stream=streams[i])
        except Exception as e:  # noqa: BLE001 - reported below
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for i in range(2):
        for k, o in enumerate(outs[i]):
            hip.assert_bit_equal(o.cpu().numpy(), refs[i], "thread %d launch %d" % (i, k))
    op.close()


ROLES_CASES = [
    replace(C.CONFIG3_SMALL, bs=3, dst_dt=C.U8),                                       # headline shape, 2 channel groups
    replace(C.CONFIG3_SMALL, bs=3, dst_dt=C.U8, wide=True),                            # saturation on both stages
    replace(C.CONFIG3_SMALL, bs=2, dst_dt=C.S8, relu1=False, wide=True),               # negative outputs, signed saturation
    replace(C.CONFIG3_SMALL, bs=2, dst_dt=C.U8, per_channel0=True, per_channel1=True),  # per-channel constants from LDS
    replace(C.CONFIG3_SMALL, bs=2, dst_dt=C.U8, bia0_dt=C.UNDEF, bia1_dt=C.UNDEF),
    replace(C.CONFIG3_SMALL, bs=2, dst_dt=C.U8, bia0_dt=C.S8, bia1_dt=C.U8),
    replace(C.SMALL64, dst_dt=C.U8),                                                   # 12x10: ragged tiles, one channel group
    replace(C.SMALL64, dst_dt=C.S8, wide=True),
    C.ConvCase("r32_384", 2, 32, 17, 23, 32, 384, dst_dt=C.U8),                         # ic = oc = 32, three channel groups
    C.ConvCase("r32_512", 1, 32, 9, 40, 32, 512, dst_dt=C.U8, pad=(0, 0)),              # four groups (two B waves serve two), no padding
    C.ConvCase("r64_32_128", 2, 64, 8, 70, 32, 128, dst_dt=C.U8, wide=True),           # ic 64 -> oc 32, column-split units
    C.ConvCase("r32_64_256", 2, 32, 30, 33, 64, 256, dst_dt=C.S8, relu1=True),
    C.ConvCase("one128", 1, 32, 3, 3, 32, 128, pad=(0, 0), dst_dt=C.U8),               # a single output pixel
]


@pytest.mark.parametrize("case", ROLES_CASES, ids=lambda c: c.ident())
def test_role_specialised_kernel(hip, oracle, tuning, case):
    """conv_mfma_roles.cuh (A waves: conv0 + requant 0 -> mid ring in LDS; B waves: conv1 + requant 1 + stores):
    bit-exact against the oracle, and byte-identical to conv_mfma.cuh's kernel (DFX_NO_ROLES=1)."""
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.kernel_name.decode().startswith("conv_mfma_roles_kernel"), info.kernel_name
    assert info.block in (768, 1024)
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "roles " + case.ident())
    tuning.setenv("DFX_NO_ROLES", "1")
    got_old, info_old = hip.hip_conv(case, data)
    assert info_old.kernel_name.decode().startswith("conv_mfma_fused_kernel"), info_old.kernel_name
    hip.assert_bit_equal(got_old, got, "roles vs resident kernel")


@pytest.mark.parametrize("geom,rounds", [("1,56", None), ("2,56", "0"), ("3,56", "1"), ("7,32", None), ("4,32", "2"), ("4,56", "1")])
def test_role_specialised_kernel_scheduling(hip, oracle, tuning, geom, rounds):
    """unit geometries (full-width linear units, column-split units with partial tiles) x unit hand-out (static
    split, device queue from the first unit on) for the role-specialised kernel; enough images that every
    loader works through several units and the mid ring wraps many times."""
    tuning.setenv("DFX_FORCE_GEOM", geom)
    if rounds is not None:
        tuning.setenv("DFX_STATIC_ROUNDS", rounds)
    case = replace(C.CONFIG3_SMALL, bs=24, dst_dt=C.U8, wide=True)
    data = C.generate(case)
    got, info = hip.hip_conv(case, data)
    assert info.kernel_name.decode().startswith("conv_mfma_roles_kernel") and info.rows_per_unit == int(geom.split(",")[0])
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "roles geom %s rounds %s" % (geom, rounds))


def test_role_specialised_kernel_s8_routes(hip, oracle):
    """s8 output on the role-specialised kernel: with ReLU the u8 route + a clamp to [0, 127] (the saturation must
    be reached in the wide cases), without ReLU the signed emitter; partial tiles (13 x 17 pixels) in both."""
    saw_127 = saw_negative = False
    for relu1 in (True, False):
        for wide in (False, True):
            for pc1 in (False, True):
                case = C.ConvCase("s8r", 3, 64, 13, 17, 64, 256, dst_dt=C.S8, relu1=relu1, wide=wide, per_channel1=pc1)
                data = C.generate(case)
                if wide:
                    data["scales1"] = data["scales1"] * np.float32(8.0)   # push the outputs into the saturation
                got, info = hip.hip_conv(case, data)
                assert info.kernel_name.decode().startswith("conv_mfma_roles_kernel"), info.kernel_name
                hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "roles s8 relu1=%s wide=%s" % (relu1, wide))
                saw_127 |= relu1 and bool((got == 127).any())
                saw_negative |= (not relu1) and bool((got < 0).any())
    assert saw_127 and saw_negative


def test_role_specialised_kernel_falls_back(hip, oracle, tuning):
    """what the role-specialised kernel does not take stays on conv_mfma.cuh's kernel: round-down, a negative
    conv0 scale, the exact-requant switches."""
    base = replace(C.CONFIG3_SMALL, bs=2, dst_dt=C.U8)
    for case, sw in ((replace(base, rm0=1), None), (base, "DFX_NO_MAGIC"), (base, "DFX_NO_FAST")):
        if sw:
            tuning.setenv(sw, "1")
        data = C.generate(case)
        got, info = hip.hip_conv(case, data)
        assert info.kernel_name.decode().startswith("conv_mfma_fused_kernel"), (info.kernel_name, sw)
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "fallback")
        if sw:
            tuning.setenv(sw, None)
    # shapes the role-specialised kernel is not built for (oc = 64 with oc1x1 = 384 / 512): create must succeed
    # and serve them with conv_mfma.cuh's kernel (dfx_conv_create failed for them until late in round 3)
    for ic, oc1 in ((64, 384), (32, 512), (64, 512)):
        for dt in (C.U8, C.S8):
            case = C.ConvCase("noroles", 2, ic, 13, 17, 64, oc1, dst_dt=dt)
            data = C.generate(case)
            got, info = hip.hip_conv(case, data)
            assert info.kernel_name.decode().startswith("conv_mfma_fused_kernel"), info.kernel_name
            hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "no roles instance")
    # ... and the largest ones it is built for
    for ic, oc, oc1 in ((64, 64, 256), (32, 64, 256), (64, 32, 512), (32, 32, 384)):
        case = C.ConvCase("roles", 2, ic, 13, 17, oc, oc1, dst_dt=C.U8)
        data = C.generate(case)
        got, info = hip.hip_conv(case, data)
        assert info.kernel_name.decode().startswith("conv_mfma_roles_kernel"), info.kernel_name
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "roles instance")
    data = C.generate(base)
    data["scales0"] = -data["scales0"]          # negative scale: the fma mode's sign argument does not hold
    got, info = hip.hip_conv(base, data)
    assert info.kernel_name.decode().startswith("conv_mfma_fused_kernel"), info.kernel_name
    hip.assert_bit_equal(got, hip.oracle_conv(oracle, base, data), "negative scale")


@pytest.mark.parametrize("geom", ["1,56", "2,56", "3,56", "4,56", "3,32", "4,32", "2,32", "7,32"])
def test_unit_geometries_and_tile_rotation(hip, oracle, geom, tuning):
    """every unit decomposition the host may pick (full-width linear units, 32-multiple
    column units, tile counts that do not divide the 7 compute waves) gives the same bytes."""
    tuning.setenv("DFX_FORCE_GEOM", geom)
    for case in (replace(C.CONFIG3_SMALL, bs=3), replace(C.CONFIG3_SMALL, bs=3, dst_dt=C.U8, wide=True)):
        data = C.generate(case)
        got, info = hip.hip_conv(case, data)
        assert info.variant == hip.dfa.VARIANT_MFMA_FUSED and info.rows_per_unit == int(geom.split(",")[0])
        hip.assert_bit_equal(got, hip.oracle_conv(oracle, case, data), "geom " + geom)
