#!/usr/bin/env python3
"""The u8-output res2a block under both stage-1 requant routes of the role-specialised kernel: bench.py's synthetic
conv1 scale happens to be a power of two (60 / (6 * 40 * sqrt(64)) = 2^-5), which admits the one-instruction "fma"
route; an arbitrary scale takes the "magic" route (v_add_f32 + v_mul_f32).  Times both, checks 8 images of each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
from dataclasses import replace
import cases as C
import hipref
from oracle import oracle as orc

case = C.ConvCase("res2a", 128, 64, 56, 56, 64, 256, dst_dt=C.U8)
base = C.generate(case)


def timed(op, srcs, dsts, steps=300):
    for i in range(60):
        op.submit(srcs[i % 4], dsts[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        op.submit(srcs[i % 4], dsts[i % 4])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


srcs = [torch.from_numpy(np.roll(base["src"], i, axis=0)).cuda() for i in range(4)]
for tag, f0, f1, pc in (("bench scales (conv1 scale 2^-5)", 1.0, 1.0, False), ("conv1 scale x 1.37", 1.0, 1.37, False),
                        ("both scales x 1.37 / 0.83", 0.83, 1.37, False), ("per-channel conv1 scales", 1.0, 1.0, True)):
    c = replace(case, per_channel1=pc)
    data = C.generate(c)
    data["scales0"] = (data["scales0"] * np.float32(f0)).astype(np.float32)
    data["scales1"] = (data["scales1"] * np.float32(f1)).astype(np.float32)
    op = hipref.make_conv(c, data)
    dsts = [torch.empty(op.dst_shape, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for _ in range(3):
        timed(op, srcs, dsts, 100)
    us = min(timed(op, srcs, dsts) for _ in range(3))
    ref = hipref.oracle_conv(orc, replace(c, bs=8), dict(data, src=data["src"][:8]))
    hipref.assert_bit_equal(dsts[0][:8].cpu().numpy(), ref, tag)
    print("%-34s %-36s %7.2f us  (8 images bit-exact)" % (tag, op.info().kernel_name.decode(), us), flush=True)
    op.close()
