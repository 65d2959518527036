"""debug helper: many launches of one case; which elements ever differ from the oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import cases as C, hipref, refmath
from oracle import oracle as orc
import test_gpu_parity as T
name = sys.argv[1]
case = [c for c in T.FUSED_CASES if c.ident() == name][0]
data = C.generate(case)
ref = hipref.oracle_conv(orc, case, data)
ref2 = refmath.conv_ref(case, data)
print("oracle == independent formulation:", np.array_equal(ref, ref2))
tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
nbad = 0
for fresh in range(int(sys.argv[2])):
    op = hipref.make_conv(case, data)
    src = torch.from_numpy(data["src"]).cuda()
    for rep in range(20):
        dst = torch.empty(op.dst_shape, dtype=tdt, device="cuda")
        dst.view(torch.uint8).fill_(0xCD)
        op.submit(src, dst)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        bad = np.argwhere(got != ref)
        if len(bad):
            nbad += 1
            print("op", fresh, "rep", rep, "bad", len(bad), [tuple(int(v) for v in b) for b in bad[:6]],
                  [(int(got[tuple(b)]), int(ref[tuple(b)])) for b in bad[:6]])
    op.close()
print("launches with mismatches:", nbad)
