#!/usr/bin/env python3
"""Diagnostic only: DFX_STAMPS build on a workload served by conv_direct.cuh; prints where a wave's cycles go."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
capi = importlib.import_module("deep-fusion_amd.capi")
capi._LIB = os.path.join(ROOT, "deep-fusion_amd", "libdfx_hip_stamps.so")
import torch
import cases as C
import hipref
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "res3"
case, desc = bench.workloads()[wl]
data = C.generate(case)
op = hipref.make_conv(case, data, force_variant=3)
info = op.info()
src = torch.from_numpy(data["src"]).cuda()
tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
out = torch.empty(op.dst_shape, dtype=tdt, device="cuda")
for _ in range(3):
    op.submit(src, out)
torch.cuda.synchronize()
L = capi.lib()
L.dfx_debug_read_stamps.restype = ctypes.c_int
L.dfx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
nw = info.block // 64
buf = np.zeros(info.grid * nw * 16, dtype=np.uint64)
n = L.dfx_debug_read_stamps(op._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
p = buf[:n].reshape(info.grid, nw, 16).astype(np.float64)
print(wl, desc, "kernel", info.kernel_name.decode(), "grid", info.grid, "lds", info.lds_bytes)
units = p[..., 7]
names = ["barrier + tile staging", "conv0 K loop", "requant 0", "barrier after conv0", "conv1 K loop", "requant 1 + stores", "whole unit"]
for k, nm in enumerate(names):
    print("%-28s %10.0f cycles/unit/wave %6.1f%%" % (nm, p[..., k].sum() / units.sum(), 100 * p[..., k].sum() / p[..., 6].sum()))
print("%-28s %10.0f cycles (mean over waves)" % ("entry -> first unit", p[..., 8].mean()))
print("%-28s %10.0f cycles (mean over waves that had units) max %.0f" % ("entry -> wave done", p[..., 9][p[..., 7] > 0].mean(), p[..., 9].max()))
