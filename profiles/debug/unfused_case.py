import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from dataclasses import replace
import cases as C
import hipref
from oracle import oracle as orc
capi = importlib.import_module("deep-fusion_amd.capi")
base = C.ConvCase(name='rnd31', bs=21, ic=192, ih=4, iw=3, oc=512, oc1x1=0, k=(3, 2), stride=(1, 2), pad=(1, 1), dst_dt=4, bia0_dt=3, bia1_dt=3, relu0=False, relu1=False, rm0=1, rm1=1, per_channel0=False, per_channel1=True, wide=False, seed=1031)
def run(case, env, tag):
    for k, v in env.items(): capi.lib().dfx_debug_set_tuning(k.encode(), v.encode())
    try:
        data = C.generate(case)
        ref = hipref.oracle_conv(orc, case, data)
        got, info = hipref.hip_conv(case, data)
        bad = np.argwhere(got != ref)
        print(tag, info.kernel_name.decode(), "grid", info.grid, "rows/unit", info.rows_per_unit, "bad", len(bad), "of", got.size,
              ("first %s; bad images %s; bad px(y,x) %s; bad ch range %d..%d" % (bad[0], sorted(set(bad[:, 0]))[:8], sorted(set(map(tuple, bad[:, 1:3])))[:8], bad[:, 3].min(), bad[:, 3].max())) if len(bad) else "")
    finally:
        for k in env: capi.lib().dfx_debug_set_tuning(k.encode(), None)
E8 = {"DFX_DIRECT_NW": "8", "DFX_DIRECT_NPB": "1"}
run(base, E8, "orig")
run(base, {"DFX_DIRECT_NW": "4", "DFX_DIRECT_NPB": "1"}, "nw4npb1")
run(base, {"DFX_DIRECT_NW": "8", "DFX_DIRECT_NPB": "2"}, "nw8npb2")
run(replace(base, k=(3, 3), pad=(1, 1)), E8, "k33")
run(replace(base, stride=(1, 1)), E8, "s11")
run(replace(base, rm0=0), E8, "rm0=0")
run(replace(base, relu0=True), E8, "relu")
run(replace(base, bs=4), E8, "bs4")
run(replace(base, oc=256), E8, "oc256")
run(replace(base, ic=128), E8, "ic128")
run(replace(base, dst_dt=C.S32), E8, "s32")
run(replace(base, oc1x1=128), E8, "fused")
