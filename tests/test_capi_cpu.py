"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol
include/dfx.h declares, validates descriptors like the reference's init_conf, and
refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import importlib

import numpy as np
import pytest

dfa = importlib.import_module("deep-fusion_amd")


def test_library_exports_every_declared_symbol():
    L = dfa.lib()
    syms = dfa.declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    assert L.dfx_version() == 100


def test_reorder_matches_oracle_layout(oracle):
    rng = np.random.default_rng(0)
    w = rng.integers(-128, 128, (48, 32, 3, 3)).astype(np.int8)
    assert np.array_equal(dfa.reorder_oihw_to_blocked(w), oracle.reorder_oihw_to_blocked(w))
    w1 = rng.integers(-128, 128, (64, 48, 1, 1)).astype(np.int8)
    assert np.array_equal(dfa.reorder_oihw_to_blocked(w1), oracle.reorder_oihw_to_blocked(w1))
    L = dfa.lib()
    assert L.dfx_blocked_offset(17, 5, 2, 1, 32, 3, 3) == ((1 * 2 + 0) * 9 + 7) * 256 + 1 * 64 + 1 * 4 + 1


def _expect_invalid(**kw):
    import torch
    args = dict(src_shape_nhwc=(1, 8, 8, 32), wei_shape_oihw=(32, 32, 3, 3), oc1x1=32)
    args.update(kw)
    with pytest.raises(dfa.DfxError) as e:
        dfa.Conv(**args)
    return str(e.value)


def test_descriptor_validation_like_init_conf():
    # reference rules: op_conv.cc:286-346, jit_conv_kernel.cc:586-592, :662-671
    assert "dfx error 1" in _expect_invalid(wei_shape_oihw=(24, 32, 3, 3))        # oc % 16
    assert "dfx error 1" in _expect_invalid(src_shape_nhwc=(1, 8, 8, 24), wei_shape_oihw=(32, 24, 3, 3))
    assert "dfx error 1" in _expect_invalid(oc1x1=40)
    assert "dfx error 1" in _expect_invalid(nscales0=7)
    assert "dfx error 1" in _expect_invalid(nscales1=5)
    assert "dfx error 1" in _expect_invalid(dst_dt=9)
    assert "channel" in _expect_invalid(wei_shape_oihw=(32, 16, 3, 3))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dfa.DfxError) as e:
        dfa.Conv((1, 8, 8, 32), (32, 32, 3, 3), oc1x1=32)
    assert "no HIP device" in str(e.value)
    with pytest.raises(dfa.DfxError):
        dfa.Concat(1, 2, 2, [16, 16], np.uint8)


def test_ctypes_structs_match_the_header(tmp_path):
    """ABI drift guard: the descriptor structs of include/dfx.h, compiled by gcc, have the sizes and field
    offsets the ctypes mirrors in capi.py assume (a field added on one side only would make the library read
    garbage behind the caller's struct)."""
    import os
    import subprocess
    capi = importlib.import_module("deep-fusion_amd.capi")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = {"dfx_conv_desc": capi.ConvDesc, "dfx_conv_info": capi.ConvInfo, "dfx_concat_desc": capi.ConcatDesc,
             "dfx_pool_desc": capi.PoolDesc, "dfx_eltwise_desc": capi.EltwiseDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dfx.h"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in ct._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0; }")
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    seen = {}
    for ln in subprocess.check_output([str(exe)]).decode().splitlines():
        a, b, c = ln.split()
        seen[(a, b)] = int(c)
    for cname, ct in pairs.items():
        assert seen[(cname, "size")] == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert seen[(cname, fname)] == getattr(ct, fname).offset, (cname, fname)
