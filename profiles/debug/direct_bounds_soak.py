"""Seeded random shapes through the bounds-checking build of conv_direct.cuh (make dkdebug): every global access
of the kernel is range-checked against its buffer (tile loads, weight rings, constants, both store paths); prints
any tag that went out of range.  usage: python profiles/debug/direct_bounds_soak.py [n_small] [n_big] [seed]"""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
capi = importlib.import_module("deep-fusion_amd.capi")
capi._LIB = os.path.join(ROOT, "deep-fusion_amd", "libdfx_hip_dkdebug.so")
import cases as C
import hipref
import test_gpu_parity as T
from oracle import oracle as orc
n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 4711
L = capi.lib()
L.dfx_debug_read_bounds.restype = ctypes.c_int
L.dfx_debug_read_bounds.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
TD = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}
bad = ran = 0
for case in T._random_cases(n_small, seed) + T._random_cases(n_big, seed + 1, big=True):
    data = C.generate(case)
    op = hipref.make_conv(case, data)
    name = op.info().kernel_name.decode()
    if not name.startswith("conv_direct_kernel"):
        op.close()
        continue
    src = torch.from_numpy(data["src"]).cuda()
    dst = torch.zeros(op.dst_shape, dtype=TD[case.dst_dt], device="cuda")
    op.submit(src, dst)
    torch.cuda.synchronize()
    out = np.zeros(64, dtype=np.int64)
    L.dfx_debug_read_bounds(op._h, out.ctypes.data_as(ctypes.c_void_p))
    hits = [(t, int(out[2 * t]), int(out[2 * t + 1])) for t in range(32) if out[2 * t] != -1 or out[2 * t + 1] != -1]
    hipref.assert_bit_equal(dst.cpu().numpy(), hipref.oracle_conv(orc, case, data), name)
    ran += 1
    if hits:
        bad += 1
        print("OUT OF RANGE", name, case, hits)
    op.close()
print("%d conv_direct cases through the bounds-checking build, %d with an out-of-range access; outputs bit-exact" % (ran, bad))
