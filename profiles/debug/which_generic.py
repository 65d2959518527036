#!/usr/bin/env python3
"""Lists which of soak_random.py's "huge" cases are served by the scalar backstop kernel, and why (create-time only)."""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cases as C
import hipref
seed, n_huge = int(sys.argv[1]), int(sys.argv[2])
hr = np.random.default_rng(seed + 2)
while n_huge > 0:
    kh, kw = int(hr.integers(1, 4)), int(hr.integers(1, 4))
    sh, sw = int(hr.integers(1, 3)), int(hr.integers(1, 3))
    ph, pw = int(hr.integers(0, kh)), int(hr.integers(0, kw))
    ih, iw = int(hr.integers(1, 10)), int(hr.integers(1, 10))
    if ih + 2 * ph < kh or iw + 2 * pw < kw:
        continue
    fused = bool(hr.integers(0, 4))
    case = C.ConvCase(
        "huge%d" % n_huge, int(hr.integers(1, 7)), 16 * int(hr.integers(1, 65)), ih, iw, 16 * int(hr.integers(1, 65)),
        16 * int(hr.integers(1, 161)) if fused else 0, k=(kh, kw), stride=(sh, sw), pad=(ph, pw),
        dst_dt=int(hr.choice([C.U8, C.S8, C.S32, C.F32])), bia0_dt=int(hr.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
        bia1_dt=int(hr.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])), relu0=bool(hr.integers(0, 2)), relu1=bool(hr.integers(0, 2)),
        rm0=int(hr.integers(0, 2)), rm1=int(hr.integers(0, 2)), per_channel0=bool(hr.integers(0, 2)),
        per_channel1=bool(hr.integers(0, 2)), wide=bool(hr.integers(0, 2)), seed=9000 + n_huge)
    n_huge -= 1
    data = C.generate(case)
    op = hipref.make_conv(case, data)
    nm = op.info().kernel_name.decode()
    op.close()
    if "generic" in nm:
        print("%-28s bs %d ic %4d %dx%d oc %4d oc1 %4d k%r s%r p%r dst %d" % (nm, case.bs, case.ic, case.ih, case.iw, case.oc, case.oc1x1, case.k, case.stride, case.pad, case.dst_dt))
