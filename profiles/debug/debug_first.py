"""debug helper: run ONE case once in a fresh process (first launch of its kernel)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases as C, hipref, refmath
from oracle import oracle as orc
import test_gpu_parity as T
name = sys.argv[1]
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for c in T.FUSED_CASES[:pre]:          # optionally replay the tests that precede it under pytest
    d = C.generate(c); hipref.hip_conv(c, d)
case = [c for c in T.FUSED_CASES if c.ident() == name][0]
data = C.generate(case)
got, info = hipref.hip_conv(case, data)
ref = hipref.oracle_conv(orc, case, data)
bad = np.argwhere(got != ref)
if len(bad):
    ref2 = refmath.conv_ref(case, data)
    print("MISMATCH", len(bad), [tuple(int(v) for v in b) for b in bad[:4]], "gpu", [int(got[tuple(b)]) for b in bad[:4]],
          "oracle", [int(ref[tuple(b)]) for b in bad[:4]], "indep", [int(ref2[tuple(b)]) for b in bad[:4]])
else:
    print("ok")
