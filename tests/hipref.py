"""Helpers shared by the GPU parity tests: run a ConvCase through the C ABI
(libdfx_hip.so via deep-fusion_amd.capi) and through the CPU oracle."""
import importlib

import numpy as np

import cases as C

dfa = importlib.import_module("deep-fusion_amd")


def oracle_conv(orc, case, data, impl=None):
    if impl is None:
        impl = "avx512" if orc.have_avx512_vnni() else "scalar_mt"
    wb = orc.reorder_oihw_to_blocked(data["w0"])
    w1b = orc.reorder_oihw_to_blocked(data["w1"]) if case.oc1x1 else None
    return orc.conv(data["src"], wb, data["w0"].shape, case.stride, case.pad, case.dst_dt,
                    data["scales0"], bia0=data["bia0"], wei1_blk=w1b, oc1x1=case.oc1x1,
                    scales1=data["scales1"], bia1=data["bia1"], relu0=case.relu0,
                    relu1=case.relu1, rm0=case.rm0, rm1=case.rm1, impl=impl)


def make_conv(case, data, force_variant=-1, fuse_pool=0):
    op = dfa.Conv(data["src"].shape, data["w0"].shape, stride=case.stride, pad=case.pad, fuse_pool=fuse_pool,
                  dst_dt=case.dst_dt, oc1x1=case.oc1x1, bia0_dt=case.bia0_dt,
                  bia1_dt=case.bia1_dt if case.oc1x1 else 0, conv0_relu=case.relu0,
                  conv1_relu=case.relu1, rm0=case.rm0, rm1=case.rm1,
                  nscales0=data["scales0"].size, nscales1=data["scales1"].size,
                  force_variant=force_variant)
    wb = dfa.reorder_oihw_to_blocked(data["w0"])
    w1b = dfa.reorder_oihw_to_blocked(data["w1"]) if case.oc1x1 else None
    op.set_weights(wb, data["scales0"], bia0=data["bia0"], wei1_blk=w1b,
                   scales1=data["scales1"] if case.oc1x1 else None, bia1=data["bia1"])
    return op


def hip_conv(case, data, force_variant=-1, host_path=False):
    """-> (dst ndarray, ConvInfo).  Device-resident path unless host_path."""
    import torch
    op = make_conv(case, data, force_variant)
    info = op.info()
    if host_path:
        out = op.submit_host(data["src"])
    else:
        src = torch.from_numpy(data["src"]).cuda()
        tdt = {C.F32: torch.float32, C.S32: torch.int32, C.S8: torch.int8, C.U8: torch.uint8}[case.dst_dt]
        dst = torch.empty(op.dst_shape, dtype=tdt, device="cuda")
        dst.view(torch.uint8).fill_(0xCD)      # poison: unwritten elements must show
        op.submit(src, dst)
        torch.cuda.synchronize()
        out = dst.cpu().numpy()
    op.close()
    return out, info


def assert_bit_equal(got, ref, what=""):
    assert got.dtype == ref.dtype and got.shape == ref.shape, (what, got.dtype, ref.dtype, got.shape, ref.shape)
    g = got.view(np.uint32) if got.dtype == np.float32 else got
    r = ref.view(np.uint32) if ref.dtype == np.float32 else ref
    if not np.array_equal(g, r):
        bad = np.argwhere(g != r)
        i = tuple(bad[0])
        spread = ["axis %d: %s" % (ax, sorted(set(int(v) for v in bad[:, ax]))[:40]) for ax in range(bad.shape[1])]
        raise AssertionError("%s: %d of %d elements differ; first at %s: got %r want %r; indices hit per axis: %s" %
                             (what, len(bad), g.size, i, got[i], ref[i], "; ".join(spread)))
