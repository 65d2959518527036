"""debug helper: scribble over every CU's LDS before each conv launch; count mismatches."""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import cases as C, hipref
from oracle import oracle as orc
import test_gpu_parity as T
S = ctypes.CDLL(os.path.join(ROOT, "deep-fusion_amd/tools/probe/liblds_scribble.so"))
name = sys.argv[1]
case = [c for c in T.FUSED_CASES if c.ident() == name][0]
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
op = hipref.make_conv(case, data)
src = torch.from_numpy(data["src"]).cuda()
nbad = 0
from collections import Counter
cnt = Counter()
for rep in range(int(sys.argv[2])):
    dst = torch.empty(op.dst_shape, dtype=tdt, device="cuda")
    dst.view(torch.uint8).fill_(0xCD)
    if len(sys.argv) > 3 and sys.argv[3] == "scribble":
        S.dbg_scribble(ctypes.c_uint(0x9E3779B9 * (rep + 1) & 0xffffffff), ctypes.c_void_p(0))
    if len(sys.argv) > 4:
        import time
        torch.cuda.synchronize(); time.sleep(float(sys.argv[4]))
    op.submit(src, dst)
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    bad = np.argwhere(got != ref)
    if len(bad):
        nbad += 1
        for b in bad:
            cnt[(int(b[2]) , int(b[3]) % 4)] += 1
        if nbad <= 6:
            print("rep", rep, "bad", len(bad), [tuple(int(v) for v in b) for b in bad[:5]])
print("launches with mismatches:", nbad, "of", sys.argv[2])
print("(col, channel%4) histogram:", sorted(cnt.items())[:40])
