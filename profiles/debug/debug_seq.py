"""debug helper: replay the FUSED_CASES sequence like pytest does, many times; for a
mismatch report GPU vs oracle vs the independent formulation and a second GPU run."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases as C, hipref, refmath
from oracle import oracle as orc
import test_gpu_parity as T
rounds = int(sys.argv[1])
nb = 0
for r in range(rounds):
    for case in T.FUSED_CASES[:40]:
        data = C.generate(case)
        got, info = hipref.hip_conv(case, data)
        ref = hipref.oracle_conv(orc, case, data)
        if not np.array_equal(got.view(np.uint8), ref.view(np.uint8)):
            nb += 1
            ref2 = refmath.conv_ref(case, data)
            got2, _ = hipref.hip_conv(case, data)
            ref3 = hipref.oracle_conv(orc, case, data, impl="scalar_mt")
            bad = np.argwhere(got != ref)
            print("round", r, case.ident(), "bad", len(bad), [tuple(int(v) for v in b) for b in bad[:3]],
                  "gpu", [got[tuple(b)] for b in bad[:3]], "gpu2", [got2[tuple(b)] for b in bad[:3]],
                  "avx512", [ref[tuple(b)] for b in bad[:3]], "scalar", [ref3[tuple(b)] for b in bad[:3]],
                  "indep", [ref2[tuple(b)] for b in bad[:3]], flush=True)
print("mismatching runs:", nb)
