#!/usr/bin/env python3
"""One-off soak (not part of the suite): seeded random cases for what the other two soaks do not reach --
concat (+relu), pooling (max / avg), eltwise-sum, convs with fused 2x2/2 pooling, pointwise unfused convs
(conv_pw.cuh) and many-channel unfused convs (conv_direct.cuh's unfused form) -- every output bit-compared with
the oracle.  usage: python profiles/debug/soak_ops.py [n per kind] [seed]"""
import os, sys, importlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import cases as C
import hipref
from test_oracle import concat_inputs, pool_input
from oracle import oracle as orc
dfa = hipref.dfa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 99
rng = np.random.default_rng(seed)
t0 = time.time()
TD = {np.float32: torch.float32, np.int32: torch.int32, np.int8: torch.int8, np.uint8: torch.uint8}


def dev_out(shape, np_dt):
    d = torch.empty(tuple(shape), dtype=TD[np_dt], device="cuda")
    d.view(torch.uint8).fill_(0xCD)
    return d


# ---- concat ----
for i in range(n):
    np_dt = [np.float32, np.int32, np.int8, np.uint8][int(rng.integers(0, 4))]
    gran = 16 if np_dt in (np.int8, np.uint8) else 4
    k = int(rng.integers(1, 9))
    nb, h, w = int(rng.integers(1, 7)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
    chans = [gran * int(rng.integers(1, 20)) for _ in range(k)]
    shapes = [(nb, c, h, w) for c in chans]
    srcs = concat_inputs(shapes, np_dt, bool(rng.integers(0, 2)), seed=int(rng.integers(0, 1 << 30)))
    relu = bool(rng.integers(0, 2))
    op = dfa.Concat(nb, h, w, chans, np_dt, relu)
    ref = orc.concat(srcs, relu)
    dst = dev_out(op.dst_shape, np_dt)
    op.submit([torch.from_numpy(s).cuda() for s in srcs], dst)
    torch.cuda.synchronize()
    hipref.assert_bit_equal(dst.cpu().numpy(), ref, "concat %r %s relu=%s" % (shapes, np_dt, relu))
    op.close()
print("progress: concat %d cases ok, %.0f s" % (n, time.time() - t0), flush=True)

# ---- pooling + eltwise ----
for i in range(n):
    np_dt = [np.uint8, np.int8, np.int32, np.float32][int(rng.integers(0, 4))]
    nb, h, w, c = int(rng.integers(1, 5)), int(rng.integers(1, 30)), int(rng.integers(1, 30)), int(rng.integers(1, 70))
    kh, kw = int(rng.integers(1, min(h, 4) + 1)), int(rng.integers(1, min(w, 4) + 1))
    sh, sw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    ph, pw = int(rng.integers(0, kh)), int(rng.integers(0, kw))
    oh, ow = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    # (the caller gives the output size; ceil-mode-like windows may hang over the bottom / right edge)
    if rng.integers(0, 2) and (oh * sh - ph) < h: oh += 1
    if rng.integers(0, 2) and (ow * sw - pw) < w: ow += 1
    x = pool_input((nb, h, w, c), np_dt, seed=int(rng.integers(0, 1 << 30)))
    src = torch.from_numpy(x).cuda()
    for algo, fn in ((dfa.Pool.MAX if hasattr(dfa.Pool, "MAX") else 0, lambda: orc.maxpool(x, (kh, kw), (sh, sw), (ph, pw), (oh, ow))),
                     (dfa.Pool.AVG_INCLUDE_PADDING, lambda: orc.avgpool(x, (kh, kw), (sh, sw), (ph, pw), (oh, ow), True)),
                     (dfa.Pool.AVG_EXCLUDE_PADDING, lambda: orc.avgpool(x, (kh, kw), (sh, sw), (ph, pw), (oh, ow), False))):
        try:
            op = dfa.Pool(nb, c, h, w, oh, ow, (kh, kw), (sh, sw), (ph, pw), np_dt, algo=algo)
        except dfa.DfxError:
            continue   # a window entirely in the padding: refused by create, as documented
        dst = dev_out(op.dst_shape, np_dt)
        op.submit(src, dst)
        torch.cuda.synchronize()
        hipref.assert_bit_equal(dst.cpu().numpy(), fn(), "pool algo %d %r k%r s%r p%r o%r %s" % (algo, x.shape, (kh, kw), (sh, sw), (ph, pw), (oh, ow), np_dt))
        op.close()
    m = int(rng.integers(2, 9))
    xs = [pool_input((nb, h, w, c), np_dt, seed=int(rng.integers(0, 1 << 30))) for _ in range(m)]
    relu = bool(rng.integers(0, 2))
    op = dfa.EltwiseSum(m, xs[0].size, np_dt, relu)
    dst = dev_out((nb, h, w, c), np_dt)
    op.submit([torch.from_numpy(v).cuda() for v in xs], dst)
    torch.cuda.synchronize()
    hipref.assert_bit_equal(dst.cpu().numpy(), orc.eltwise_sum(xs, relu), "eltwise %r x%d %s" % (xs[0].shape, m, np_dt))
    op.close()
print("progress: pool/eltwise %d cases ok, %.0f s" % (n, time.time() - t0), flush=True)

# ---- conv + fused 2x2/2 max pooling (resident-weight kernel) ----
done = 0
for i in range(n):
    ih, iw = 2 * int(rng.integers(1, 20)), 2 * int(rng.integers(1, 60))
    pad = int(rng.integers(0, 2))
    if pad == 0:
        ih, iw = ih + 2, iw + 2
    case = C.ConvCase("pf%d" % i, int(rng.integers(1, 6)), int(rng.choice([32, 64])), ih, iw, int(rng.choice([32, 64])), 0,
                      pad=(pad, pad), dst_dt=int(rng.choice([C.U8, C.S8, C.S32, C.F32])),
                      bia0_dt=int(rng.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])),
                      relu0=bool(rng.integers(0, 2)), rm0=int(rng.integers(0, 2)),
                      per_channel0=bool(rng.integers(0, 2)), wide=bool(rng.integers(0, 2)), seed=7000 + i)
    data = C.generate(case)
    mid = hipref.oracle_conv(orc, case, data)
    ref = orc.maxpool(mid, (2, 2), (2, 2), (0, 0), (mid.shape[1] // 2, mid.shape[2] // 2))
    op = hipref.make_conv(case, data, fuse_pool=2)
    dst = dev_out(op.dst_shape, C.NP_OF[case.dst_dt])
    op.submit(torch.from_numpy(data["src"]).cuda(), dst)
    torch.cuda.synchronize()
    hipref.assert_bit_equal(dst.cpu().numpy(), ref, "fused pool %r" % (case,))
    op.close()
    done += 1
print("progress: fused pool %d cases ok, %.0f s" % (done, time.time() - t0), flush=True)

# ---- pointwise unfused (conv_pw.cuh) and many-channel unfused (conv_direct.cuh) ----
kernels = {}
for i in range(n):
    if i % 2 == 0:   # pointwise: ic % 256 == 0, oc in {64,128,256}, oc * ic <= 98304
        ic = 256 * int(rng.integers(1, 7))
        ocs = [o for o in (64, 128, 256) if o * ic <= 98304]
        case = C.ConvCase("pw%d" % i, int(rng.integers(1, 9)), ic, int(rng.integers(1, 30)), int(rng.integers(1, 30)),
                          int(rng.choice(ocs)), 0, k=(1, 1), pad=(0, 0))
    else:            # unfused, >= 64 channels, any window
        kh, kw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        case = C.ConvCase("uf%d" % i, int(rng.integers(1, 9)), 64 * int(rng.integers(1, 9)), int(rng.integers(3, 20)), int(rng.integers(3, 20)),
                          64 * int(rng.integers(1, 9)), 0, k=(kh, kw), stride=(int(rng.integers(1, 3)), int(rng.integers(1, 3))),
                          pad=(int(rng.integers(0, kh)), int(rng.integers(0, kw))))
    from dataclasses import replace
    case = replace(case, dst_dt=int(rng.choice([C.U8, C.S8, C.S32, C.F32])),
                   bia0_dt=int(rng.choice([C.UNDEF, C.S8, C.U8, C.S32, C.F32])), relu0=bool(rng.integers(0, 2)),
                   rm0=int(rng.integers(0, 2)), per_channel0=bool(rng.integers(0, 2)), wide=bool(rng.integers(0, 2)), seed=8000 + i)
    data = C.generate(case)
    ref = hipref.oracle_conv(orc, case, data)
    got, info = hipref.hip_conv(case, data)
    nm = info.kernel_name.decode()
    key = nm.split("<")[0] + ("/unfused" if "unfused" in nm else "")
    kernels[key] = kernels.get(key, 0) + 1
    hipref.assert_bit_equal(got, ref, "%s %r" % (nm, case))
print("all kinds bit-exact; unfused kernels: %s; %.0f s" % (kernels, time.time() - t0))
