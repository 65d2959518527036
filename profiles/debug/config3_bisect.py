"""diagnostic: config 3 (N=128, s32) mismatch count per library switch, a few launches each"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import cases as C, hipref
from dataclasses import replace
from oracle import oracle as orc
capi = importlib.import_module("deep-fusion_amd.capi")
if os.environ.get("DFX_LIB_PATH"):
    print("library:", os.environ["DFX_LIB_PATH"], flush=True)
orc.build()
case = replace(C.CONFIG3_SMALL, bs=32)
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
big = replace(case, bs=128)
src128 = np.concatenate([data["src"]] * 4)
ref128 = np.concatenate([ref] * 4)
d128 = dict(data, src=src128)
for sw in sys.argv[1:] or ["none"]:
    for kv in sw.split(","):
        if kv != "none":
            k, v = kv.split("=")
            capi.set_tuning(k, v)
    op = hipref.make_conv(big, d128)
    src = torch.from_numpy(src128).cuda()
    dst = torch.empty(op.dst_shape, dtype=torch.int32, device="cuda")
    counts = []
    for rep in range(4):
        dst.fill_(-7)
        op.submit(src, dst)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        bad = np.argwhere(got != ref128)
        counts.append(len(bad))
        if len(bad) and rep == 0:
            vals = sorted(set(int(got[tuple(b)]) for b in bad[:200]))[:8]
            print("   e.g.", [tuple(int(v) for v in b) for b in bad[:4]], "values", vals, "channels", sorted(set(int(b[3]) for b in bad))[:24], flush=True)
    print(sw, op.info().kernel_name.decode(), "rows/unit", op.info().rows_per_unit, "mismatches per launch:", counts, flush=True)
    op.close()
    for kv in sw.split(","):
        if kv != "none":
            capi.set_tuning(kv.split("=")[0], None)
