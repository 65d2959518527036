"""debug helper: run one named test case repeatedly and report mismatches."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases as C, hipref
from oracle import oracle as orc
import test_gpu_parity as T
name = sys.argv[1]
case = [c for c in T.FUSED_CASES if c.ident() == name][0]
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    got, info = hipref.hip_conv(case, data)
    bad = np.argwhere(got != ref)
    print(trial, info.kernel_name.decode(), "grid", info.grid, "rows", info.rows_per_unit, "lds", info.lds_bytes,
          "bad", len(bad), [tuple(b) for b in bad[:4]], [(got[tuple(b)], ref[tuple(b)]) for b in bad[:4]])
