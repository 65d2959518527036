"""The built HIP library must keep 2 wait states between every 16-byte global store and the first
VALU write to its data registers (gfx950 needs 2, hipcc pads 1: tools/probe/probe_store_war.hip,
DESIGN.md section 4.1).  Checked on the disassembly, no GPU needed."""
import importlib
import os

import pytest

import isa_scan

capi = importlib.import_module("deep-fusion_amd.capi")


@pytest.mark.skipif(not os.path.exists(isa_scan.OBJDUMP), reason="llvm-objdump not available")
def test_no_valu_overwrite_within_two_wait_states_of_a_16_byte_store():
    lib = capi.lib_path()
    assert os.path.exists(lib), "libdfx_hip.so has not been built"
    total, bad = 0, []
    objs = isa_scan.code_objects(lib)
    assert objs, "no gfx950 code object found in " + lib
    for o in objs:
        n, b = isa_scan.scan_object(o)
        total += n
        bad += b
    assert total > 100, total          # the s32 / f32 / staged-u8 epilogues and concat are 16-byte stores
    assert not bad, "%d hazards, e.g. %s" % (len(bad), bad[:3])


@pytest.mark.skipif(not os.path.exists(isa_scan.OBJDUMP), reason="llvm-objdump not available")
def test_no_packed_f32_arithmetic_in_the_library():
    """Round 3: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 do not overlap with MFMAs on gfx950 (in one wave's
    stream each adds ~10 cycles to an MFMA-paced loop, beside another wave's MFMA stream they run at 13.8 cycles
    each; plain v_add_f32 / v_mul_f32 / v_fma_f32 / v_cvt_pk_u8_f32 hide under the MFMAs:
    tools/probe/probe_coexec.hip, profiles/r03/probe_coexec.jsonl).  The requant arithmetic is written with
    scalar float operations and the library is built with -fno-slp-vectorize; no packed-f32 instruction may be
    left.  (This also retires round 2's op_sel hazard -- `op_sel:[0,1]` forms returned wrong low lanes,
    tools/probe/probe_pk_opsel.hip -- there is no instruction left that could take that form.)"""
    lib = capi.lib_path()
    forms = None
    for o in isa_scan.code_objects(lib):
        c = isa_scan.packed_f32_opsel_forms(o)
        forms = c if forms is None else forms + c
    assert forms is not None, "no gfx950 code object found in " + lib
    assert not forms, dict(forms)
