"""debug helper: where do GPU and oracle outputs differ (per image/row/col histograms)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases as C, hipref
from dataclasses import replace
from oracle import oracle as orc
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
case = replace(C.CONFIG3_SMALL, name="cfg3", bs=bs)
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
for trial in range(3):
    got, info = hipref.hip_conv(case, data)
    bad = (got != ref).any(axis=3)
    print("trial", trial, info.kernel_name.decode(), "rows/unit", info.rows_per_unit, "grid", info.grid,
          "bad px", int(bad.sum()), "of", bad.size)
    if bad.sum():
        n, y, x = np.nonzero(bad)
        print(" images with errors:", len(np.unique(n)), "first", np.unique(n)[:20])
        print(" rows hist:", np.bincount(y, minlength=56))
        print(" cols hist:", np.bincount(x, minlength=56))
        i = (n[0], y[0], x[0])
        d = np.nonzero(got[i] != ref[i])[0]
        print(" first bad px", i, "channels wrong", len(d), d[:16], "got", got[i][d[:6]], "ref", ref[i][d[:6]])
        # is the wrong pixel equal to some other pixel's reference? (misplaced tile)
        flat = ref.reshape(-1, ref.shape[3])
        m = np.nonzero((flat == got[i]).all(axis=1))[0]
        print(" matches ref of pixel index:", m[:5], "own index", (i[0]*56+i[1])*56+i[2])
