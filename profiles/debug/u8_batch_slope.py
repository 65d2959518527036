"""Launch time of the u8-output res2a block against the batch size: the intercept is the kernel's fill / drain,
the slope its steady-state rate.  Beside it: a pure write stream of the same number of bytes (torch fill_) and a
device copy of half as many (read + write = the same traffic).  python profiles/debug/u8_batch_slope.py [dst]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
from dataclasses import replace
import cases as C
import hipref

dst_name = sys.argv[1] if len(sys.argv) > 1 else "u8"
DT = {"u8": C.U8, "s8": C.S8, "s32": C.S32, "f32": C.F32}[dst_name]
TDT = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[DT]
base = C.ConvCase("res2a", 128, 64, 56, 56, 64, 256, dst_dt=DT)


def timed(fn, steps=200, warm=40):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3  # us


rows = []
data128 = C.generate(base)
for n in (16, 32, 64, 128, 256, 384, 512):
    case = replace(base, bs=n)
    reps = (n + 127) // 128
    src = np.concatenate([data128["src"]] * reps, axis=0)[:n]
    data = dict(data128, src=src)
    op = hipref.make_conv(case, data)
    info = op.info()
    nbuf = 4
    srcs = [torch.from_numpy(np.roll(src, i, axis=0)).cuda() for i in range(nbuf)]
    dsts = [torch.empty(op.dst_shape, dtype=TDT, device="cuda") for _ in range(nbuf)]
    # ramp the clocks
    for _ in range(3):
        timed(lambda i: op.submit(srcs[i % nbuf], dsts[i % nbuf]), steps=100, warm=10)
    us = min(timed(lambda i: op.submit(srcs[i % nbuf], dsts[i % nbuf])) for _ in range(3))
    byts = int(info.algorithmic_bytes)
    nbytes = dsts[0].numel() * dsts[0].element_size()
    fill_us = min(timed(lambda i: dsts[i % nbuf].view(torch.uint8).fill_(7)) for _ in range(3))
    rows.append((n, us, byts, fill_us, nbytes))
    print("N=%4d  %-44s grid %4d  %8.2f us  %7.1f MB  %6.2f TB/s | fill_ of the %6.1f MB dst: %7.2f us  %5.2f TB/s"
          % (n, info.kernel_name.decode(), info.grid, us, byts / 1e6, byts / us / 1e6, nbytes / 1e6, fill_us,
             nbytes / fill_us / 1e6), flush=True)
    op.close()
    del srcs, dsts
    torch.cuda.empty_cache()

print()
for (n0, t0, b0, f0, _), (n1, t1, b1, f1, _) in zip(rows, rows[1:]):
    print("N %4d -> %4d: +%7.2f us for +%6.1f MB = %5.2f TB/s marginal (fill_: +%6.2f us)"
          % (n0, n1, t1 - t0, (b1 - b0) / 1e6, (b1 - b0) / (t1 - t0) / 1e6, f1 - f0))
xs = np.array([r[2] for r in rows if r[0] >= 128], dtype=np.float64)
ys = np.array([r[1] for r in rows if r[0] >= 128], dtype=np.float64)
k, c = np.polyfit(xs, ys, 1)
print("fit over N >= 128: %.2f us + bytes / %.2f TB/s" % (c, 1.0 / k / 1e6))
