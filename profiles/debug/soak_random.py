#!/usr/bin/env python3
"""One-off soak (not part of the suite): many more seeded random shapes than tests/test_gpu_parity.py runs, through the
default dispatch and through randomly forced conv_direct.cuh forms; every output bit-compared with the oracle.
usage: python profiles/debug/soak_random.py [n_small] [n_big] [seed] [n_huge]   (huge: up to 1024 / 1024 / 2560 channels)"""
import os, sys, importlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cases as C
import hipref
import test_gpu_parity as T
from oracle import oracle as orc
capi = importlib.import_module("deep-fusion_amd.capi")
n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31337
rng = np.random.default_rng(seed)
n_huge = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cases = T._random_cases(n_small, seed) + T._random_cases(n_big, seed + 1, big=True)
hr = np.random.default_rng(seed + 2)
while n_huge > 0:
    kh, kw = int(hr.integers(1, 4)), int(hr.integers(1, 4))
    sh, sw = int(hr.integers(1, 3)), int(hr.integers(1, 3))
    ph, pw = int(hr.integers(0, kh)), int(hr.integers(0, kw))
    ih, iw = int(hr.integers(1, 10)), int(hr.integers(1, 10))
    if ih + 2 * ph < kh or iw + 2 * pw < kw:
        continue
    fused = bool(hr.integers(0, 4))
    cases.append(C.ConvCase(
        "huge%d" % n_huge, int(hr.integers(1, 7)), 16 * int(hr.integers(1, 65)), ih, iw, 16 * int(hr.integers(1, 65)),
        16 * int(hr.integers(1, 161)) if fused else 0, k=(kh, kw), stride=(sh, sw), pad=(ph, pw),
        dst_dt=int(hr.choice([C.U8, C.S8, C.S32, C.F32])), bia0_dt=int(hr.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
        bia1_dt=int(hr.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])), relu0=bool(hr.integers(0, 2)), relu1=bool(hr.integers(0, 2)),
        rm0=int(hr.integers(0, 2)), rm1=int(hr.integers(0, 2)), per_channel0=bool(hr.integers(0, 2)),
        per_channel1=bool(hr.integers(0, 2)), wide=bool(hr.integers(0, 2)), seed=9000 + n_huge))
    n_huge -= 1
kernels = {}
t0 = time.time()
for i, case in enumerate(cases):
    data = C.generate(case)
    ref = hipref.oracle_conv(orc, case, data)
    forced = {}
    if rng.integers(0, 2):
        forced = {"DFX_DIRECT_NW": str(rng.choice([4, 8])), "DFX_DIRECT_NPB": str(rng.choice([1, 2, 4]))}
    for k, v in forced.items():
        capi.lib().dfx_debug_set_tuning(k.encode(), v.encode())
    try:
        got, info = hipref.hip_conv(case, data)
    finally:
        for k in forced:
            capi.lib().dfx_debug_set_tuning(k.encode(), None)
    name = info.kernel_name.decode().split("<")[0] + ("/unfused" if "unfused" in info.kernel_name.decode() else "")
    kernels[name] = kernels.get(name, 0) + 1
    hipref.assert_bit_equal(got, ref, "%s %r forced %r" % (info.kernel_name.decode(), case, forced))
    if i % 50 == 49:
        print("progress: %d / %d cases, %.0f s" % (i + 1, len(cases), time.time() - t0), flush=True)
print("all %d cases bit-exact; kernels: %s" % (len(cases), kernels))
