#!/usr/bin/env python3
"""Diagnostic only: runs the DFX_STAMPS build (libdfx_hip_stamps.so) of the role-specialised fused kernel
(conv_mfma_roles.cuh) on the res2a block and prints, per role, where a wave's cycles go.  Never quote this
build's run time; read its shares.   usage: stamps_roles.py [u8|s8]"""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
capi = importlib.import_module("deep-fusion_amd.capi")
capi._LIB = os.environ.get("DFX_STAMPS_LIB") or os.path.join(ROOT, "deep-fusion_amd", "libdfx_hip_stamps.so")
import torch  # noqa: E402
import cases as C  # noqa: E402
import hipref  # noqa: E402

dst = {"u8": C.U8, "s8": C.S8}[sys.argv[1] if len(sys.argv) > 1 else "u8"]
case = C.ConvCase("res2a", 128, 64, 56, 56, 64, 256, dst_dt=dst)
data = C.generate(case)
op = hipref.make_conv(case, data)
info = op.info()
src = torch.from_numpy(data["src"]).cuda()
out = torch.empty(op.dst_shape, dtype=torch.uint8 if dst == C.U8 else torch.int8, device="cuda")
for _ in range(5):
    op.submit(src, out)
torch.cuda.synchronize()
L = capi.lib()
L.dfx_debug_read_stamps.restype = ctypes.c_int
L.dfx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(info.grid * 768, dtype=np.uint64)
L.dfx_debug_read_stamps(op._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
p = buf[:info.grid * 256].reshape(info.grid, 16, 16).astype(np.float64)
p = p[:, p[0, :, 13] > 0, :]   # the waves that exist (12 or 16)
role = p[..., 15]
print("kernel", info.kernel_name.decode(), "grid", info.grid, "rows/unit", info.rows_per_unit)
life = p[..., 13] - p[..., 12]
rt = p[..., 14]
mhz = None
span_cyc = p[..., 13].max() - p[..., 12].min()
print("wave lifetime (cycles): mean %.0f min %.0f max %.0f; first entry -> last exit %.0f cycles; exit spread %.2f us (s_memrealtime)" % (
    life.mean(), life.min(), life.max(), span_cyc, (rt.max() - rt.min()) / 100.0))
for r, name, cols in ((0, "A (conv0 + requant 0)", ["wait for the tile's unit", "address math + conv0", "requant 0", "tiles", "entry -> loop", "wait for a free mid slot", "mid write + publish"]),
                      (1, "B (conv1 + requant 1 + stores)", ["wait for a mid slot", "conv1 + requant 1 + store issue", "-", "tiles", "entry -> loop"]),
                      (2, "loader", ["wait for the slot", "tile write + publish", "draw + prefetch issue", "units", "entry -> loop"])):
    w = p[role == r]
    if not len(w):
        continue
    n = np.maximum(w[:, 3], 1)
    print("== %s: %d waves, %.2f %s per wave (min %.0f max %.0f), lifetime %.0f cycles" % (
        name, len(w), w[:, 3].mean(), "units" if r == 2 else "tiles", w[:, 3].min(), w[:, 3].max(), (w[:, 13] - w[:, 12]).mean()))
    for k, c in enumerate(cols):
        if c in ("-", "tiles", "units"):
            continue
        if c == "entry -> loop":
            print("   %-34s %8.0f cycles" % (c, w[:, k].mean()))
            print("   start-up: global loads arrived %6.0f, own staging done (at the barrier) %6.0f, past the barrier %6.0f (cycles after entry); entry stamp spread over the workgroup's waves %6.0f cycles" % (
                w[:, 7].mean(), w[:, 8].mean(), w[:, 9].mean(), (p[..., 12].max(axis=1) - p[..., 12].min(axis=1)).mean()))
        else:
            print("   %-34s %8.0f cycles per %s   (%.1f%% of the wave's lifetime)" % (
                c, (w[:, k] / n).mean(), "unit" if r == 2 else "tile", 100 * (w[:, k] / (w[:, 13] - w[:, 12])).mean()))
