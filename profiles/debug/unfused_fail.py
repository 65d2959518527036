"""debug: where does the unfused 120x360 case differ (positions, lanes), under the mode switches"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import cases as C, hipref
from oracle import oracle as orc
capi = importlib.import_module("deep-fusion_amd.capi")
orc.build()
case = C.unfused(C.REF_SHAPES[2])
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
for sw in (None, "DFX_NO_MAGIC", "DFX_NO_FAST"):
    if sw: capi.set_tuning(sw, "1")
    for rep in range(2):
        got, info = hipref.hip_conv(case, data)
        bad = np.argwhere(got != ref)
        print(sw, "rep", rep, info.kernel_name.decode(), "rows/unit", info.rows_per_unit, "bad", len(bad),
              [tuple(int(v) for v in b) for b in bad[:20]], [(int(got[tuple(b)]), int(ref[tuple(b)])) for b in bad[:6]], flush=True)
    if sw: capi.set_tuning(sw, None)
