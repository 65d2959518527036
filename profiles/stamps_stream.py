#!/usr/bin/env python3
"""Diagnostic only: runs the DFX_STAMPS build on a bench workload served by the
streamed-weight kernel (conv_stream.cuh) and prints where a wave's cycles go.
Never quote this build's run time; read its shares."""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
capi = importlib.import_module("deep-fusion_amd.capi")
capi._LIB = os.path.join(ROOT, "deep-fusion_amd", "libdfx_hip_stamps.so")
import torch  # noqa: E402
import cases as C  # noqa: E402
import hipref  # noqa: E402
import bench  # noqa: E402
from dataclasses import replace  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "res3"
case, desc = bench.workloads()[wl]
if len(sys.argv) > 2:
    case = replace(case, dst_dt={"u8": C.U8, "s32": C.S32, "f32": C.F32, "s8": C.S8}[sys.argv[2]])
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
buf = np.zeros(info.grid * 96, dtype=np.uint64)
n = L.dfx_debug_read_stamps(op._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
p = buf[:n].reshape(info.grid, 4, 24).astype(np.float64)
print(wl, desc, "kernel", info.kernel_name.decode(), "grid", info.grid, "lds", info.lds_bytes)
units = p[..., 8]
print("units per workgroup: mean %.2f min %.0f max %.0f" % (units.mean(), units.min(), units.max()))
names = ["tile staging (+barrier)", "conv0 phases (steps)", "conv1 steps", "-",
         "-", "conv0 epilogue", "conv1 epilogue + stores", "whole unit"]
tot = p[..., 7].sum()
for k, nm in enumerate(names):
    if nm == '-':
        continue
    print("%-36s %10.0f cycles/unit/wave %6.1f%%" % (nm, p[..., k].sum() / units.sum(), 100 * p[..., k].sum() / tot))
life = p[..., 9]
print("wave lifetime: mean %.0f min %.0f max %.0f cycles (s_memtime ~100 MHz x21?)" % (life.mean(), life.min(), life.max()))
steps = p[..., 19].sum()
for k, nm in zip(range(10, 19), ["MFMA set0 first 2", "issue LDS loads set1", "MFMA set0 rest", "MFMA set1 first 2 (waits set1)",
                                 "issue LDS loads next set0", "weights commit + issue", "tile prefetch issue",
                                 "MFMA set1 rest", "barrier"]):
    print("conv0 step: %-34s %8.0f cycles/step" % (nm, p[..., k].sum() / steps))
