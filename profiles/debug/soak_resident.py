#!/usr/bin/env python3
"""One-off soak (not part of the suite) for the resident-weight kernels (conv_mfma.cuh, conv_mfma_roles.cuh):
seeded random shapes inside their domain (3x3, stride 1, pad 0/1, 32/64 channels, fused or not), sized so that the
loaders go through static rounds AND the queue, every output bit-compared with the oracle.
usage: python profiles/debug/soak_resident.py [n] [seed] [roles]   (roles: only shapes the role-specialised kernel may take)"""
import os, sys, importlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cases as C
import hipref
from oracle import oracle as orc
capi = importlib.import_module("deep-fusion_amd.capi")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
roles = len(sys.argv) > 3 and sys.argv[3] == "roles"
rng = np.random.default_rng(seed)
kernels, t0 = {}, time.time()
for i in range(n):
    big = i % 3 == 0   # every third case: many units per loader (queue draws)
    ih, iw = (int(rng.integers(20, 97)), int(rng.integers(20, 97))) if big else (int(rng.integers(1, 40)), int(rng.integers(1, 70)))
    bs = int(rng.integers(8, 49)) if big else int(rng.integers(1, 9))
    ph, pw = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    if ih + 2 * ph < 3 or iw + 2 * pw < 3:
        continue
    fused = roles or bool(rng.integers(0, 4))
    oc = int(rng.choice([32, 64]))
    case = C.ConvCase("res%d" % i, bs, int(rng.choice([32, 64])), ih, iw, oc,
                      (128 * int(rng.integers(1, 3 if oc == 64 else 5)) if roles else 32 * int(rng.integers(1, 17))) if fused else 0,
                      pad=(ph, pw),
                      dst_dt=int(rng.choice([C.U8, C.S8] if roles else [C.U8, C.U8, C.S8, C.S32, C.F32])),
                      bia0_dt=int(rng.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
                      bia1_dt=int(rng.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
                      relu0=bool(rng.integers(0, 2)), relu1=bool(rng.integers(0, 2)),
                      rm0=0 if roles else int(rng.integers(0, 2)), rm1=int(rng.integers(0, 2)),
                      per_channel0=bool(rng.integers(0, 2)), per_channel1=bool(rng.integers(0, 2)),
                      wide=bool(rng.integers(0, 2)), seed=5000 + i)
    data = C.generate(case)
    ref = hipref.oracle_conv(orc, case, data)
    got, info = hipref.hip_conv(case, data)
    name = info.kernel_name.decode().split("<")[0]
    kernels[name] = kernels.get(name, 0) + 1
    hipref.assert_bit_equal(got, ref, "%s %r" % (info.kernel_name.decode(), case))
    if i % 20 == 19:
        print("progress: %d / %d cases, %.0f s" % (i + 1, n, time.time() - t0), flush=True)
print("all cases bit-exact; kernels: %s" % kernels)
