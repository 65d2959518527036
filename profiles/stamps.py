#!/usr/bin/env python3
"""Diagnostic only: runs the DFX_STAMPS build (libdfx_hip_stamps.so: in-kernel
s_memtime stamps) on a bench workload and prints where a workgroup's cycles go.
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
from dataclasses import replace  # noqa: E402

dst = {"u8": C.U8, "s32": C.S32, "f32": C.F32, "s8": C.S8}[sys.argv[1] if len(sys.argv) > 1 else "u8"]
case = C.ConvCase("res2a", 128, 64, 56, 56, 64, 256, dst_dt=dst)
data = C.generate(case)
op = hipref.make_conv(case, data)
info = op.info()
src = torch.from_numpy(data["src"]).cuda()
tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[dst]
out = torch.empty(op.dst_shape, dtype=tdt, device="cuda")
for _ in range(3):
    op.submit(src, out)
torch.cuda.synchronize()
L = capi.lib()
L.dfx_debug_read_stamps.restype = ctypes.c_int
L.dfx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(info.grid * 768, dtype=np.uint64)
# the kernel overwrites (not accumulates across launches): values are of the last launch
n = L.dfx_debug_read_stamps(op._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size)
p = buf[:info.grid * 256].reshape(info.grid, 2, 8, 16).astype(np.float64)
tl = buf[info.grid * 256:info.grid * 768].reshape(info.grid, 16, 8, 4)
comp = p[:, :, :7, :]
print("kernel", info.kernel_name.decode(), "grid", info.grid, "rows/unit", info.rows_per_unit)
# slots (conv_mfma.cuh, DFX_ACC): 0 claim + wait for the unit's tile, 1 conv0, 2 requant0, 3 conv1 + requant1 + stores,
# 4 start-up, 5 whole claim, 6 tiles computed, 7 claims made
print("tiles per compute wave: mean %.2f (min %.0f max %.0f); claims per wave: %.2f" % (
    comp[..., 6].mean(), comp[..., 6].min(), comp[..., 6].max(), comp[..., 7].mean()))
print("start-up (entry -> claim loop): mean %.0f cycles, of which this wave's weight staging (loads + LDS writes) %.0f; rest = barrier wait" % (comp[..., 4].mean(), comp[..., 12].mean()))
life = comp[..., 9] - comp[..., 8]
ld = p[:, :, 7, :]
lu = max(ld[..., 3].sum(), 1.0)
print("loader waves, per unit staged by the loader itself (%.1f per loader): wait for the slot %.0f cycles, tile write + publish %.0f, next draw + prefetch issue %.0f" % (
    ld[..., 3].mean(), ld[..., 0].sum() / lu, ld[..., 1].sum() / lu, ld[..., 2].sum() / lu))
print("loader waves: entry -> barrier %.0f cycles (max %.0f), in the barrier %.0f" % (ld[..., 13].mean(), ld[..., 13].max(), ld[..., 14].mean()))
print("wave lifetime entry->exit: mean %.0f  min %.0f  max %.0f cycles" % (life.mean(), life.min(), life.max()))
rt = comp[..., 10]
rte = comp[..., 11]
print("s_memtime ticks per s_memrealtime tick (10 ns): %.2f  => s_memtime runs at %.0f MHz" % ((life / (rt - rte)).mean(), (life / (rt - rte)).mean() * 100))
print("kernel span first entry -> last exit: %.2f us; exit time spread over compute waves %.2f us" % ((rt.max() - rte.min()) / 100.0, (rt.max() - rt.min()) / 100.0))
ex = (rt.max() - rt.reshape(info.grid, -1).max(axis=1)) / 100.0   # per CU: idle us before the kernel ends
print("CU idle time before kernel end (us): mean %.2f  percentiles 10/50/90/100: %s  => %.1f%% of the span" % (ex.mean(), [round(float(v), 2) for v in np.percentile(ex, [10, 50, 90, 100])], 100 * ex.mean() / ((rt.max() - rte.min()) / 100.0)))
wex = (rt.reshape(info.grid, -1).max(axis=1, keepdims=True) - rt.reshape(info.grid, -1)) / 100.0
print("wave idle before its own CU's last wave exits (us): mean %.2f max %.2f" % (wex.mean(), wex.max()))
tiles = comp[..., 6].sum()
names = ["claim + wait for tile", "conv0 MFMA", "requant0", "conv1+requant1+stores"]
tot = comp[..., 5].sum()
for k, nm in enumerate(names):
    print("compute waves  %-24s %10.0f cycles/tile  %5.1f%% of the claim loop" % (nm, comp[..., k].sum() / tiles, 100 * comp[..., k].sum() / tot))
print("compute waves  %-24s %10.0f cycles/tile" % ("whole loop", tot / tiles))
wg_life = life.reshape(info.grid, -1).max(axis=1)
print("per-XCD (blockIdx % 8) mean WG lifetime:", [int(wg_life[x::8].mean()) for x in range(8)])
q = np.percentile(wg_life, [0, 10, 50, 90, 100])
print("WG lifetime percentiles 0/10/50/90/100:", [int(v) for v in q])


def timeline(cu):
    """per-wave event log of one workgroup: compute waves (claim start, tile start, tile end, tile id), loaders"""
    print("-- timeline of workgroup %d (cycles from each wave's kernel entry)" % cu)
    for w in range(16):
        ev = tl[cu, w]
        loader = w in (7, 15)
        items = []
        for e in ev:
            if not e.any():
                continue
            if loader:
                items.append("k%d[wait %d..%d write ..%d prefetch ..%d]" % (int(e[3]) & 0xffff, int(e[0]), int(e[1]), int(e[2]), int(e[3]) >> 16))
            else:
                items.append("t%d%s[%d +%d +%d]" % (int(e[3]) >> 8, "" if int(e[3]) & 1 else "x", int(e[0]), int(e[1] - e[0]), int(e[2] - e[1])))
        print(("L" if loader else "c") + "%02d " % w + " ".join(items))


wg_order = np.argsort(wg_life)
timeline(int(wg_order[len(wg_order) // 2]))
timeline(int(wg_order[-1]))
