"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package (deep-fusion_amd/) never does.  See dfo_oracle.h for
the reference file:line each function restates and for the parity status
("parity unpinned" for conv, pinned for concat).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdfx_oracle.so")

# deepfusion::memory::dtype values (reference include/deepfusion.h:66-72)
UNDEF, F32, S32, S8, U8 = 0, 1, 2, 3, 4
ROUND_NEAREST, ROUND_DOWN = 0, 1

NP_OF = {F32: np.float32, S32: np.int32, S8: np.int8, U8: np.uint8}
DT_OF = {np.dtype(np.float32): F32, np.dtype(np.int32): S32,
         np.dtype(np.int8): S8, np.dtype(np.uint8): U8}


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "bs", "ic", "ih", "iw", "oc", "oh", "ow", "kh", "kw", "sh", "sw", "pt", "pl",
        "oc1x1", "dst_dt", "bia0_dt", "bia1_dt", "relu0", "relu1", "rm0", "rm1",
        "nscale0", "nscale1")]


def build(force=False):
    """Compile oracle/libdfx_oracle.so with the committed Makefile."""
    # always through make (a no-op when the library is newer than its sources): a stale library
    # after a source change must not pass for the oracle
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp = ctypes.c_void_p
        for name in ("dfo_conv_scalar", "dfo_conv_scalar_mt", "dfo_conv_avx512"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
        L.dfo_concat.restype = ctypes.c_int
        L.dfo_concat.argtypes = [ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int),
                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, vp]
        L.dfo_maxpool.restype = ctypes.c_int
        L.dfo_maxpool.argtypes = [vp, vp] + [ctypes.c_int] * 13
        L.dfo_avgpool.restype = ctypes.c_int
        L.dfo_avgpool.argtypes = [vp, vp] + [ctypes.c_int] * 14
        L.dfo_eltwise_sum.restype = ctypes.c_int
        L.dfo_eltwise_sum.argtypes = [ctypes.c_int, ctypes.POINTER(vp), vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
        L.dfo_reorder_oihw_to_blocked.restype = None
        L.dfo_reorder_oihw_to_blocked.argtypes = [vp, vp] + [ctypes.c_int] * 4
        L.dfo_have_avx512_vnni.restype = ctypes.c_int
        L.dfo_num_threads.restype = ctypes.c_int
        L.dfo_set_num_threads.restype = None
        L.dfo_set_num_threads.argtypes = [ctypes.c_int]
        L.dfo_conv_out_size.restype = ctypes.c_int
        L.dfo_conv_out_size.argtypes = [ctypes.c_int] * 4
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def reorder_oihw_to_blocked(w_oihw):
    """plain oihw s8 -> OIhw4i16o4i (flat s8 array of the same size)."""
    w = np.ascontiguousarray(w_oihw, dtype=np.int8)
    O, I, KH, KW = w.shape
    out = np.empty(w.size, dtype=np.int8)
    lib().dfo_reorder_oihw_to_blocked(_ptr(w), _ptr(out), O, I, KH, KW)
    return out


def conv_out_size(image, kernel, stride, padding):
    return (image + 2 * padding - kernel) // stride + 1


def make_desc(src, wei_shape, stride, pad, dst_dt, oc1x1=0, bia0=None, bia1=None,
              relu0=False, relu1=False, rm0=ROUND_NEAREST, rm1=ROUND_NEAREST,
              nscale0=1, nscale1=1):
    bs, ih, iw, ic = src.shape
    oc, ic2, kh, kw = wei_shape
    assert ic2 == ic
    d = ConvDesc()
    d.bs, d.ic, d.ih, d.iw = bs, ic, ih, iw
    d.oc, d.kh, d.kw = oc, kh, kw
    d.sh, d.sw = stride
    d.pt, d.pl = pad
    d.oh = conv_out_size(ih, kh, d.sh, d.pt)
    d.ow = conv_out_size(iw, kw, d.sw, d.pl)
    d.oc1x1 = oc1x1
    d.dst_dt = dst_dt
    d.bia0_dt = UNDEF if bia0 is None else DT_OF[bia0.dtype]
    d.bia1_dt = UNDEF if bia1 is None else DT_OF[bia1.dtype]
    d.relu0, d.relu1, d.rm0, d.rm1 = int(relu0), int(relu1), rm0, rm1
    d.nscale0, d.nscale1 = nscale0, nscale1
    return d


def conv(src, wei_blk, wei_shape, stride, pad, dst_dt, scales0, bia0=None,
         wei1_blk=None, oc1x1=0, scales1=None, bia1=None, relu0=False, relu1=False,
         rm0=ROUND_NEAREST, rm1=ROUND_NEAREST, impl="scalar"):
    """Run the oracle.  src: NHWC u8 ndarray; wei_blk / wei1_blk: flat s8 arrays in
    OIhw4i16o4i order; returns the NHWC dst ndarray of dtype dst_dt."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    scales0 = np.ascontiguousarray(scales0, dtype=np.float32)
    if scales1 is None:
        scales1 = np.ones(1, dtype=np.float32)
    scales1 = np.ascontiguousarray(scales1, dtype=np.float32)
    d = make_desc(src, wei_shape, stride, pad, dst_dt, oc1x1, bia0, bia1, relu0, relu1,
                  rm0, rm1, scales0.size, scales1.size)
    cout = oc1x1 if oc1x1 else d.oc
    dst = np.zeros((d.bs, d.oh, d.ow, cout), dtype=NP_OF[dst_dt])
    fn = {"scalar": lib().dfo_conv_scalar, "scalar_mt": lib().dfo_conv_scalar_mt,
          "avx512": lib().dfo_conv_avx512}[impl]
    rc = fn(ctypes.byref(d), _ptr(src), _ptr(wei_blk), _ptr(bia0), _ptr(scales0),
            _ptr(wei1_blk), _ptr(bia1), _ptr(scales1), _ptr(dst))
    if rc != 0:
        raise RuntimeError("oracle conv (%s) failed with code %d" % (impl, rc))
    return dst


def concat(srcs, post_relu=False):
    """srcs: list of NHWC ndarrays of one dtype -> channel-concatenated ndarray."""
    srcs = [np.ascontiguousarray(s) for s in srcs]
    dt = DT_OF[srcs[0].dtype]
    bs, h, w, _ = srcs[0].shape
    ch = (ctypes.c_int * len(srcs))(*[s.shape[3] for s in srcs])
    ptrs = (ctypes.c_void_p * len(srcs))(*[s.ctypes.data for s in srcs])
    dst = np.zeros((bs, h, w, sum(s.shape[3] for s in srcs)), dtype=srcs[0].dtype)
    rc = lib().dfo_concat(len(srcs), ptrs, ch, bs, h, w, dt, int(post_relu), _ptr(dst))
    if rc != 0:
        raise RuntimeError("oracle concat failed with code %d" % rc)
    return dst


def maxpool(src, kernel, stride, pad, out_hw):
    """src: NHWC ndarray; max pooling over the window positions inside the input -> (bs, oh, ow, c)."""
    src = np.ascontiguousarray(src)
    bs, ih, iw, c = src.shape
    oh, ow = out_hw
    dst = np.zeros((bs, oh, ow, c), dtype=src.dtype)
    rc = lib().dfo_maxpool(_ptr(src), _ptr(dst), bs, c, ih, iw, oh, ow, kernel[0], kernel[1], stride[0], stride[1],
                           pad[0], pad[1], DT_OF[src.dtype])
    if rc != 0:
        raise RuntimeError("oracle maxpool failed with code %d" % rc)
    return dst


def avgpool(src, kernel, stride, pad, out_hw, include_padding):
    """src: NHWC ndarray; average pooling (divisor kh*kw or the number of positions inside the input)."""
    src = np.ascontiguousarray(src)
    bs, ih, iw, c = src.shape
    oh, ow = out_hw
    dst = np.zeros((bs, oh, ow, c), dtype=src.dtype)
    rc = lib().dfo_avgpool(_ptr(src), _ptr(dst), bs, c, ih, iw, oh, ow, kernel[0], kernel[1], stride[0], stride[1],
                           pad[0], pad[1], DT_OF[src.dtype], 1 if include_padding else 2)
    if rc != 0:
        raise RuntimeError("oracle avgpool failed with code %d" % rc)
    return dst


def eltwise_sum(srcs, post_relu=False):
    """srcs: list of equally shaped ndarrays of one dtype -> relu?(saturate(sum))."""
    srcs = [np.ascontiguousarray(s) for s in srcs]
    ptrs = (ctypes.c_void_p * len(srcs))(*[s.ctypes.data for s in srcs])
    dst = np.zeros_like(srcs[0])
    rc = lib().dfo_eltwise_sum(len(srcs), ptrs, _ptr(dst), srcs[0].size, DT_OF[srcs[0].dtype], int(post_relu))
    if rc != 0:
        raise RuntimeError("oracle eltwise_sum failed with code %d" % rc)
    return dst


def have_avx512_vnni():
    return bool(lib().dfo_have_avx512_vnni())


def num_threads():
    return int(lib().dfo_num_threads())


def set_num_threads(n):
    lib().dfo_set_num_threads(int(n))
