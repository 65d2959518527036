"""debug: which tiles / channel groups of the role-specialised kernel's output are wrong or unwritten"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from dataclasses import replace
import cases as C, hipref
from oracle import oracle as orc
orc.build()
case = replace(C.CONFIG3_SMALL, bs=3, dst_dt=C.U8)
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
for rep in range(2):
    got, info = hipref.hip_conv(case, data)
    bad = got != ref
    unw = (got == 0xCD) & bad
    print(info.kernel_name.decode(), "grid", info.grid, "bad", int(bad.sum()), "of which 0xCD", int(unw.sum()))
    # per (image, unit row of 4, tile of 32 px, channel group of 128)
    tiles = bad.reshape(3, 14, 7, 32, 2, 128).any(axis=(3, 5))
    print("bad (image, unit, tile, group):", [tuple(int(v) for v in t) for t in np.argwhere(tiles)][:60])
