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
def test_no_packed_f32_takes_a_high_half_for_its_low_lane():
    """v_pk_add_f32 / v_pk_mul_f32 with `op_sel:[..1..]` (low result computed from the HIGH half of a
    source pair) returned wrong low results in the last 16 lanes of a wave about once per 1e4 epilogue
    executions on gfx950 (tools/probe/probe_pk_opsel.hip; conv_mfma.cuh, mfma_cst_floats).  The
    library keeps broadcast constants as {k, k} pairs in LDS so that hipcc never needs that form."""
    lib = capi.lib_path()
    forms = None
    for o in isa_scan.code_objects(lib):
        c = isa_scan.packed_f32_opsel_forms(o)
        forms = c if forms is None else forms + c
    assert forms and sum(forms.values()) > 1000
    risky = {k: v for k, v in forms.items() if "op_sel:[" in k[1] and "1" in k[1].split("op_sel:[")[1].split("]")[0]}
    assert not risky, risky
