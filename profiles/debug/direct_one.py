"""one fused case through conv_direct.cuh vs the oracle (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
assert torch.cuda.is_available()
import cases as C
import hipref
from oracle import oracle as orc
case = C.ConvCase("res3", 2, 128, 14, 14, 128, 512, dst_dt=C.S32)
data = C.generate(case)
print("creating", flush=True)
op = hipref.make_conv(case, data, force_variant=3)
info = op.info()
print(info.kernel_name, info.grid, info.lds_bytes, flush=True)
src = torch.from_numpy(data["src"]).cuda()
dst = torch.empty(op.dst_shape, dtype=torch.int32, device="cuda")
print("submit", flush=True)
op.submit(src, dst)
torch.cuda.synchronize()
print("done", flush=True)
ref = hipref.oracle_conv(orc, case, data)
got = dst.cpu().numpy()
print("mismatches", int((got != ref).sum()), "of", ref.size)
