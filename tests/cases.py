"""Seeded test cases for the fused conv path.

Shapes: the three shapes of the reference's (empty) conv test with oh/ow
corrected (/root/reference/test/test_conv.cc:87-92), BASELINE.json
config 2, and small wide-range cases.  Value ranges follow the reference's
generators (/root/reference/test/test_utils.h:49-63: u8 in [0,16], s8/s32 in
[-10,10]) with a seeded numpy generator instead of the unseeded rand(); the
"wide" mode uses full-range u8/s8 so that saturation and s16-pair overflow
(never reached by the reference's ranges) are exercised.
"""
from dataclasses import dataclass, field, replace
from typing import Optional, Tuple

import numpy as np

UNDEF, F32, S32, S8, U8 = 0, 1, 2, 3, 4
NP_OF = {F32: np.float32, S32: np.int32, S8: np.int8, U8: np.uint8}
NAME_OF = {UNDEF: "none", F32: "f32", S32: "s32", S8: "s8", U8: "u8"}


@dataclass(frozen=True)
class ConvCase:
    name: str
    bs: int
    ic: int
    ih: int
    iw: int
    oc: int
    oc1x1: int = 0                # 0 = unfused
    k: Tuple[int, int] = (3, 3)
    stride: Tuple[int, int] = (1, 1)
    pad: Tuple[int, int] = (1, 1)
    dst_dt: int = U8
    bia0_dt: int = S32
    bia1_dt: int = S32
    relu0: bool = True
    relu1: bool = True
    rm0: int = 0
    rm1: int = 0
    per_channel0: bool = False
    per_channel1: bool = False
    wide: bool = False
    seed: int = 1234

    @property
    def oh(self):
        return (self.ih + 2 * self.pad[0] - self.k[0]) // self.stride[0] + 1

    @property
    def ow(self):
        return (self.iw + 2 * self.pad[1] - self.k[1]) // self.stride[1] + 1

    def ident(self):
        return "%s-%s-b0%s-b1%s-r%d%d-m%d%d-pc%d%d%s" % (
            self.name, NAME_OF[self.dst_dt], NAME_OF[self.bia0_dt], NAME_OF[self.bia1_dt],
            self.relu0, self.relu1, self.rm0, self.rm1, self.per_channel0, self.per_channel1,
            "-wide" if self.wide else "")


def _bias(rng, n, dt, wide):
    if dt == UNDEF:
        return None
    if dt == F32:
        return (rng.integers(-10, 11, n) + rng.random(n)).astype(np.float32)
    if dt == U8:
        return rng.integers(0, 256 if wide else 17, n).astype(np.uint8)
    if dt == S8:
        return rng.integers(-128 if wide else -10, 128 if wide else 11, n).astype(np.int8)
    lim = 5000 if wide else 10
    return rng.integers(-lim, lim + 1, n).astype(np.int32)


def generate(case: ConvCase):
    """-> dict(src NHWC u8, w0 oihw s8, w1 oihw s8|None, bia0, bia1, scales0, scales1)."""
    rng = np.random.default_rng(case.seed)
    kh, kw = case.k
    if case.wide:
        src = rng.integers(0, 256, (case.bs, case.ih, case.iw, case.ic)).astype(np.uint8)
        w0 = rng.integers(-128, 128, (case.oc, case.ic, kh, kw)).astype(np.int8)
    else:
        src = rng.integers(0, 17, (case.bs, case.ih, case.iw, case.ic)).astype(np.uint8)
        w0 = rng.integers(-10, 11, (case.oc, case.ic, kh, kw)).astype(np.int8)
    w1 = None
    if case.oc1x1:
        lo, hi = (-128, 128) if case.wide else (-10, 11)
        w1 = rng.integers(lo, hi, (case.oc1x1, case.oc, 1, 1)).astype(np.int8)
    # scales sized so that the intermediate sits inside (and sometimes beyond) u8
    k_elems = case.ic * kh * kw
    amp0 = (74.0 * 147.0 if case.wide else 6.0 * 9.0) * np.sqrt(k_elems)
    s0 = np.float32(80.0 / amp0)
    amp1 = (74.0 * 60.0 if case.wide else 6.0 * 40.0) * np.sqrt(max(case.oc, 1))
    s1 = np.float32(60.0 / amp1)
    if case.per_channel0:
        scales0 = (s0 * (0.5 + np.arange(case.oc) / case.oc)).astype(np.float32)
    else:
        scales0 = np.array([s0], dtype=np.float32)
    if case.per_channel1 and case.oc1x1:
        scales1 = (s1 * (0.5 + np.arange(case.oc1x1) / case.oc1x1)).astype(np.float32)
    else:
        scales1 = np.array([s1], dtype=np.float32)
    return dict(src=src, w0=w0, w1=w1,
                bia0=_bias(rng, case.oc, case.bia0_dt, case.wide),
                bia1=_bias(rng, case.oc1x1, case.bia1_dt, case.wide) if case.oc1x1 else None,
                scales0=scales0, scales1=scales1)


# --- shapes -----------------------------------------------------------------
# reference test/test_conv.cc:87-92 with consistent oh/ow (13x13 k3 p0 -> 11x11,
# p1 -> 13x13); the 120x360 shape is kept at batch 1 for CPU time.
REF_SHAPES = [
    ConvCase("t13p0", 2, 32, 13, 13, 32, 64, pad=(0, 0)),
    ConvCase("t13p1", 2, 32, 13, 13, 32, 32, pad=(1, 1)),
    ConvCase("t120x360", 1, 32, 120, 360, 64, 32, pad=(1, 1)),
]
CONFIG2 = ConvCase("cfg2", 1, 32, 28, 28, 32, 64)               # BASELINE.json configs[1]
CONFIG3_SMALL = ConvCase("cfg3n2", 2, 64, 56, 56, 64, 256, dst_dt=S32)  # configs[2] at N=2
CONFIG5_TINY = ConvCase("cfg5n1", 1, 64, 224, 224, 64, 128, dst_dt=F32)  # configs[4] at N=1
SMALL = ConvCase("s9x7", 2, 32, 9, 7, 32, 32)                   # ragged rows, tail tiles
SMALL64 = ConvCase("s12x10", 1, 64, 12, 10, 64, 128)

# dtype matrix of the reference test (test/test_conv.cc:95-102): bias {s8,s32} x dst {u8,s8,s32,f32}
DTYPE_MATRIX = [(b, d) for b in (S8, S32) for d in (U8, S8, S32, F32)]


def dtype_matrix(base: ConvCase):
    out = []
    for b, d in DTYPE_MATRIX:
        out.append(replace(base, bia0_dt=b, bia1_dt=b, dst_dt=d))
    return out


def option_sweep(base: ConvCase):
    """round modes, per-channel scales, relu flags, missing/f32/u8 bias, wide range."""
    out = [
        replace(base, rm0=1, rm1=1),
        replace(base, per_channel0=True, per_channel1=True),
        replace(base, relu0=False, relu1=False, dst_dt=S32),
        replace(base, relu0=False, relu1=False, dst_dt=S8),
        replace(base, relu1=False, dst_dt=F32, per_channel1=True),
        replace(base, bia0_dt=UNDEF, bia1_dt=UNDEF),
        replace(base, bia0_dt=F32, bia1_dt=U8),
        replace(base, wide=True),
        replace(base, wide=True, dst_dt=S8, relu1=False, rm1=1),
        replace(base, wide=True, dst_dt=S32, per_channel0=True),
        replace(base, wide=True, dst_dt=F32, bia1_dt=F32),
    ]
    return out


def unfused(base: ConvCase):
    return replace(base, name=base.name + "u", oc1x1=0)
