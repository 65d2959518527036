"""Regenerates tests/golden/conv_*.npz and concat_*.npz.

The reference holds no golden vectors for the conv path ("parity
unpinned") and cannot be built or imported here, so these fixtures are produced by
the independent float64/numpy formulation (tests/refmath.py) and are only written
when the C oracle (scalar and, where available, the AVX-512 instruction witness)
reproduces them bit for bit.  Inputs follow the reference generators' ranges
(/root/reference/test/test_utils.h:49-63) from a seeded numpy generator.

    python tests/golden/make_golden.py
"""
import dataclasses
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import cases as C            # noqa: E402
import refmath               # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_oracle import run_oracle, concat_inputs, CONCAT_SHAPES, CONCAT_NARROW  # noqa: E402
from dataclasses import replace  # noqa: E402

GOLDEN_CASES = [
    C.CONFIG2,
    replace(C.CONFIG2, dst_dt=C.S32, per_channel0=True, per_channel1=True, bia0_dt=C.S8, bia1_dt=C.S8),
    replace(C.SMALL, dst_dt=C.F32, relu1=False),
    replace(C.SMALL, dst_dt=C.S8, relu1=False, rm0=1, rm1=1, wide=True),
    replace(C.SMALL64, dst_dt=C.S32, wide=True, per_channel1=True),
    replace(C.REF_SHAPES[0], dst_dt=C.U8),
    replace(C.REF_SHAPES[1], dst_dt=C.S32, bia0_dt=C.UNDEF, bia1_dt=C.F32),
    C.unfused(replace(C.SMALL, dst_dt=C.S8, relu0=False)),
    C.unfused(replace(C.SMALL64, dst_dt=C.F32, relu0=True, per_channel0=True)),
    # general shapes: stride 2, 5x5, 1x1, > 64 channels, 16/48/80 channels
    C.ConvCase("s2", 1, 16, 11, 9, 48, 80, stride=(2, 2)),
    C.ConvCase("k5", 1, 16, 9, 9, 16, 16, k=(5, 5), pad=(2, 2), dst_dt=C.S32),
    C.ConvCase("k1f", 2, 48, 6, 5, 80, 48, k=(1, 1), pad=(0, 0), dst_dt=C.F32, relu1=False),
    C.ConvCase("ic128", 1, 128, 6, 6, 80, 48, dst_dt=C.S32, wide=True),
    C.ConvCase("res3", 1, 128, 7, 9, 128, 256, dst_dt=C.U8, wide=True, per_channel0=True, per_channel1=True),
]


def main():
    for i, case in enumerate(GOLDEN_CASES):
        data = C.generate(case)
        ref = refmath.conv_ref(case, data)
        for impl in ("scalar", "avx512"):
            if impl == "avx512" and not orc.have_avx512_vnni():
                continue
            got = run_oracle(orc, case, data, impl)
            assert got.dtype == ref.dtype and np.array_equal(
                got.view(np.uint8), ref.view(np.uint8)), (case, impl)
        out = {"case_" + k: np.asarray(v) for k, v in dataclasses.asdict(case).items()}
        out.update({k: v for k, v in data.items() if v is not None})
        out["dst"] = ref
        path = os.path.join(HERE, "conv_%02d_%s.npz" % (i, case.name))
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))
    # concat: shapes of reference test/test_concat.cc:122-145, all four dtypes
    for j, np_dt in enumerate((np.float32, np.int32, np.int8, np.uint8)):
        shapes = [CONCAT_SHAPES[2], CONCAT_SHAPES[4]] + ([CONCAT_NARROW[1]] if j < 2 else [])
        out = {}
        for si, sh in enumerate(shapes):
            srcs = concat_inputs(sh, np_dt, wide=True, seed=11 + si)
            for r in (0, 1):
                ref = np.concatenate(srcs, axis=3)
                if r and np_dt != np.uint8:
                    ref = np.maximum(ref, np_dt(0))
                assert np.array_equal(orc.concat(srcs, bool(r)), ref)
                out["s%d_relu%d_dst" % (si, r)] = ref
            for k, s in enumerate(srcs):
                out["s%d_src%d" % (si, k)] = s
        path = os.path.join(HERE, "concat_%s.npz" % np.dtype(np_dt).name)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
