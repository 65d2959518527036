"""deep-fusion_amd -- MI355X (gfx950) implementation of deep-fusion's hot path.

The product is the C-ABI shared library libdfx_hip.so (include/dfx.h) built from
csrc/, plus the C++ drop-in layer in host/ that re-implements the reference's
include/deepfusion.h API on top of it.  This Python package is plumbing only: it
builds the library and binds the C ABI with ctypes so tests and bench.py can
drive it with torch-owned device memory and streams.  There is no Python or CPU
compute path: every op call fails loudly when the HIP library is missing.

Import with ``importlib.import_module("deep-fusion_amd")`` (the directory name
carries a hyphen).
"""
from .capi import (  # noqa: F401
    DFX_F32, DFX_S32, DFX_S8, DFX_U8, DFX_UNDEF, ROUND_NEAREST, ROUND_DOWN,
    VARIANT_GENERIC, VARIANT_MFMA_FUSED, VARIANT_MFMA_CONV, VARIANT_MFMA_STREAM, DfxError, ConvDesc, ConvInfo, Conv, Concat, Pool, EltwiseSum,
    lib, lib_path, build, reorder_oihw_to_blocked, declared_symbols,
)
