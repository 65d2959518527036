"""ctypes binding of include/dfx.h (the drop-in C ABI).  Plumbing, not product."""
import ctypes
import os
import re
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_LIB = os.environ.get("DFX_LIB_PATH") or os.path.join(_PKG, "libdfx_hip.so")  # override: debugging only
_HEADER = os.path.join(_ROOT, "include", "dfx.h")

DFX_UNDEF, DFX_F32, DFX_S32, DFX_S8, DFX_U8 = 0, 1, 2, 3, 4
ROUND_NEAREST, ROUND_DOWN = 0, 1
VARIANT_GENERIC, VARIANT_MFMA_FUSED, VARIANT_MFMA_CONV, VARIANT_MFMA_STREAM = 0, 1, 2, 3
_NP = {DFX_F32: np.float32, DFX_S32: np.int32, DFX_S8: np.int8, DFX_U8: np.uint8}
_DT = {np.dtype(np.float32): DFX_F32, np.dtype(np.int32): DFX_S32,
       np.dtype(np.int8): DFX_S8, np.dtype(np.uint8): DFX_U8}


class DfxError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "bs", "ic", "ih", "iw", "oc", "oh", "ow", "kh", "kw", "sh", "sw", "pad_t", "pad_l",
        "oc1x1", "dst_dt", "bia0_dt", "bia1_dt", "conv0_relu", "conv1_relu",
        "conv0_round_mode", "conv1_round_mode", "conv0_nscales", "conv1_nscales",
        "force_variant", "fuse_pool")]


class ConvInfo(ctypes.Structure):
    _fields_ = [("variant", ctypes.c_int32), ("grid", ctypes.c_int32), ("block", ctypes.c_int32),
                ("lds_bytes", ctypes.c_int32), ("rows_per_unit", ctypes.c_int32),
                ("device", ctypes.c_int32), ("algorithmic_ops", ctypes.c_uint64),
                ("algorithmic_bytes", ctypes.c_uint64), ("kernel_name", ctypes.c_char * 96)]


class ConcatDesc(ctypes.Structure):
    _fields_ = [("n_inputs", ctypes.c_int32), ("bs", ctypes.c_int32), ("h", ctypes.c_int32),
                ("w", ctypes.c_int32), ("dt", ctypes.c_int32), ("post_relu", ctypes.c_int32),
                ("channels", ctypes.POINTER(ctypes.c_int32))]


class PoolDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("bs", "c", "ih", "iw", "oh", "ow", "kh", "kw", "sh", "sw",
                                             "pad_t", "pad_l", "dt", "algo")]


class EltwiseDesc(ctypes.Structure):
    _fields_ = [("n_inputs", ctypes.c_int32), ("elems", ctypes.c_int64), ("dt", ctypes.c_int32),
                ("post_relu", ctypes.c_int32)]


def lib_path():
    return _LIB


def build(force=False, jobs=8):
    """Compile csrc/ for gfx950 into libdfx_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "clean"],
                              stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-j%d" % jobs, "-s"])
    return _LIB


def declared_symbols():
    """Every function name include/dfx.h declares."""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfx_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    """Load libdfx_hip.so; raises DfxError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise DfxError("libdfx_hip.so is missing: run __graft_entry__.build() "
                       "(there is no CPU fallback for the deep-fusion hot path)")
    # PyTorch bundles its own HIP runtime under the same soname: when both live in one process
    # torch's must be loaded first (the other order leaves torch with "No HIP GPUs are
    # available").  This binding exists for tests / bench.py, which use torch for device memory.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(_LIB)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    sig = {
        "dfx_version": (i32, []),
        "dfx_last_error": (ctypes.c_char_p, []),
        "dfx_device_count": (i32, [ctypes.POINTER(i32)]),
        "dfx_set_device": (i32, [i32]),
        "dfx_device_name": (i32, [ctypes.c_char_p, sz]),
        "dfx_mem_alloc_host": (i32, [ctypes.POINTER(vp), sz]),
        "dfx_mem_free_host": (i32, [vp]),
        "dfx_mem_alloc_device": (i32, [ctypes.POINTER(vp), sz]),
        "dfx_mem_free_device": (i32, [vp]),
        "dfx_memcpy_h2d": (i32, [vp, vp, sz, vp]),
        "dfx_memcpy_d2h": (i32, [vp, vp, sz, vp]),
        "dfx_memset_device": (i32, [vp, i32, sz, vp]),
        "dfx_stream_create": (i32, [ctypes.POINTER(vp)]),
        "dfx_stream_destroy": (i32, [vp]),
        "dfx_stream_sync": (i32, [vp]),
        "dfx_stream_wait_stream": (i32, [vp, vp]),
        "dfx_event_create": (i32, [ctypes.POINTER(vp)]),
        "dfx_event_record": (i32, [vp, vp]),
        "dfx_event_elapsed_ms": (i32, [vp, vp, ctypes.POINTER(ctypes.c_float)]),
        "dfx_event_destroy": (i32, [vp]),
        "dfx_reorder_oihw_to_blocked": (i32, [vp, vp, i32, i32, i32, i32]),
        "dfx_blocked_offset": (sz, [i32] * 7),
        "dfx_conv_create": (i32, [ctypes.POINTER(ConvDesc), ctypes.POINTER(vp)]),
        "dfx_conv_set_weights": (i32, [vp] * 7),
        "dfx_conv_submit": (i32, [vp, vp, vp, vp]),
        "dfx_conv_submit_host": (i32, [vp, vp, vp]),
        "dfx_conv_query": (i32, [vp, ctypes.POINTER(ConvInfo)]),
        "dfx_conv_destroy": (i32, [vp]),
        "dfx_concat_create": (i32, [ctypes.POINTER(ConcatDesc), ctypes.POINTER(vp)]),
        "dfx_concat_submit": (i32, [vp, ctypes.POINTER(vp), vp, vp]),
        "dfx_concat_submit_host": (i32, [vp, ctypes.POINTER(vp), vp]),
        "dfx_concat_submit_gathered": (i32, [vp, vp, ctypes.POINTER(ctypes.c_uint64), vp, vp]),
        "dfx_concat_destroy": (i32, [vp]),
        "dfx_pool_create": (i32, [ctypes.POINTER(PoolDesc), ctypes.POINTER(vp)]),
        "dfx_pool_submit": (i32, [vp, vp, vp, vp]),
        "dfx_pool_destroy": (i32, [vp]),
        "dfx_eltwise_create": (i32, [ctypes.POINTER(EltwiseDesc), ctypes.POINTER(vp)]),
        "dfx_eltwise_submit": (i32, [vp, ctypes.POINTER(vp), vp, vp]),
        "dfx_eltwise_destroy": (i32, [vp]),
        "dfx_debug_scribble_lds": (i32, [ctypes.c_uint, vp]),
        "dfx_debug_set_tuning": (i32, [ctypes.c_char_p, ctypes.c_char_p]),
    }
    for name, (res, args) in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            if os.environ.get("DFX_LIB_PATH"):  # an older diagnostic build (A/B against a past round): bind what it has
                continue
            raise
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def set_tuning(key, value):
    """testing / tuning switch of the library; value None clears it."""
    _check(lib().dfx_debug_set_tuning(key.encode(), None if value is None else str(value).encode()))


def _check(rc):
    if rc != 0:
        raise DfxError("dfx error %d: %s" % (rc, lib().dfx_last_error().decode()))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def reorder_oihw_to_blocked(w_oihw):
    w = np.ascontiguousarray(w_oihw, dtype=np.int8)
    O, I, KH, KW = w.shape
    out = np.empty(w.size, dtype=np.int8)
    _check(lib().dfx_reorder_oihw_to_blocked(_p(w), _p(out), O, I, KH, KW))
    return out


def _dev_ptr(t):
    """torch CUDA tensor or raw int -> device pointer."""
    return ctypes.c_void_p(t if isinstance(t, int) else t.data_ptr())


def _stream_ptr(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream if isinstance(stream, int) else stream.cuda_stream)


class Conv:
    """dfx_conv_* handle: the op_conv<T> of the reference (src/op_conv.h:34-96)."""

    def __init__(self, src_shape_nhwc, wei_shape_oihw, stride=(1, 1), pad=(1, 1), dst_dt=DFX_U8,
                 oc1x1=0, bia0_dt=DFX_UNDEF, bia1_dt=DFX_UNDEF, conv0_relu=False,
                 conv1_relu=False, rm0=ROUND_NEAREST, rm1=ROUND_NEAREST, nscales0=1, nscales1=1,
                 force_variant=-1, fuse_pool=0):
        bs, ih, iw, ic = src_shape_nhwc
        oc, ic2, kh, kw = wei_shape_oihw
        if ic2 != ic:
            raise DfxError("Input channel do not match")
        d = ConvDesc()
        d.bs, d.ic, d.ih, d.iw, d.oc, d.kh, d.kw = bs, ic, ih, iw, oc, kh, kw
        d.sh, d.sw = stride
        d.pad_t, d.pad_l = pad
        d.oh = (ih + 2 * d.pad_t - kh) // d.sh + 1
        d.ow = (iw + 2 * d.pad_l - kw) // d.sw + 1
        d.oc1x1, d.dst_dt, d.bia0_dt, d.bia1_dt = oc1x1, dst_dt, bia0_dt, bia1_dt
        d.conv0_relu, d.conv1_relu = int(conv0_relu), int(conv1_relu)
        d.conv0_round_mode, d.conv1_round_mode = rm0, rm1
        d.conv0_nscales, d.conv1_nscales = nscales0, nscales1
        d.force_variant = force_variant
        d.fuse_pool = fuse_pool
        self.desc = d
        self._h = ctypes.c_void_p()
        _check(lib().dfx_conv_create(ctypes.byref(d), ctypes.byref(self._h)))
        self.dst_shape = ((bs, d.oh // 2, d.ow // 2, oc) if fuse_pool else (bs, d.oh, d.ow, oc1x1 if oc1x1 else oc))
        self.dst_np_dtype = _NP[dst_dt]

    def set_weights(self, wei_blk, scales0, bia0=None, wei1_blk=None, scales1=None, bia1=None):
        ws = [np.ascontiguousarray(wei_blk, dtype=np.int8),
              None if bia0 is None else np.ascontiguousarray(bia0),
              np.ascontiguousarray(scales0, dtype=np.float32),
              None if wei1_blk is None else np.ascontiguousarray(wei1_blk, dtype=np.int8),
              None if bia1 is None else np.ascontiguousarray(bia1),
              None if scales1 is None else np.ascontiguousarray(scales1, dtype=np.float32)]
        _check(lib().dfx_conv_set_weights(self._h, _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]),
                                          _p(ws[4]), _p(ws[5])))

    def submit(self, src_dev, dst_dev, stream=None):
        """asynchronous; src_dev / dst_dev are torch CUDA tensors (or raw pointers)."""
        _check(lib().dfx_conv_submit(self._h, _dev_ptr(src_dev), _dev_ptr(dst_dev),
                                     _stream_ptr(stream)))

    def submit_host(self, src_np):
        src = np.ascontiguousarray(src_np, dtype=np.uint8)
        dst = np.empty(self.dst_shape, dtype=self.dst_np_dtype)
        _check(lib().dfx_conv_submit_host(self._h, _p(src), _p(dst)))
        return dst

    def info(self):
        i = ConvInfo()
        _check(lib().dfx_conv_query(self._h, ctypes.byref(i)))
        return i

    def close(self):
        if self._h:
            lib().dfx_conv_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """dfx_pool_* handle: the pooling stage of the reference's planned conv+relu+pool op (max pooling, NHWC)."""

    MAX, AVG_INCLUDE_PADDING, AVG_EXCLUDE_PADDING = 0, 1, 2

    def __init__(self, bs, c, ih, iw, oh, ow, kernel, stride, pad, np_dtype, algo=0):
        d = PoolDesc(bs, c, ih, iw, oh, ow, kernel[0], kernel[1], stride[0], stride[1], pad[0], pad[1],
                     _DT[np.dtype(np_dtype)], algo)
        self.dst_shape = (bs, oh, ow, c)
        self._h = ctypes.c_void_p()
        _check(lib().dfx_pool_create(ctypes.byref(d), ctypes.byref(self._h)))

    def submit(self, src_dev, dst_dev, stream=None):
        _check(lib().dfx_pool_submit(self._h, _dev_ptr(src_dev), _dev_ptr(dst_dev), _stream_ptr(stream)))

    def close(self):
        if self._h:
            lib().dfx_pool_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EltwiseSum:
    """dfx_eltwise_* handle: the reference's planned eltwise-sum + relu op."""

    def __init__(self, n_inputs, elems, np_dtype, post_relu=False):
        d = EltwiseDesc(n_inputs, elems, _DT[np.dtype(np_dtype)], int(post_relu))
        self._h = ctypes.c_void_p()
        _check(lib().dfx_eltwise_create(ctypes.byref(d), ctypes.byref(self._h)))

    def submit(self, srcs_dev, dst_dev, stream=None):
        ptrs = (ctypes.c_void_p * len(srcs_dev))(*[_dev_ptr(s).value for s in srcs_dev])
        _check(lib().dfx_eltwise_submit(self._h, ptrs, _dev_ptr(dst_dev), _stream_ptr(stream)))

    def close(self):
        if self._h:
            lib().dfx_eltwise_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Concat:
    """dfx_concat_* handle: the op_concat<T> of the reference (src/op_concat.h:28-61)."""

    def __init__(self, bs, h, w, channels, np_dtype, post_relu=False):
        self.channels = list(channels)
        self._ch = (ctypes.c_int32 * len(channels))(*channels)
        d = ConcatDesc(len(channels), bs, h, w, _DT[np.dtype(np_dtype)], int(post_relu), self._ch)
        self.np_dtype = np.dtype(np_dtype)
        self.dst_shape = (bs, h, w, sum(channels))
        self._h = ctypes.c_void_p()
        _check(lib().dfx_concat_create(ctypes.byref(d), ctypes.byref(self._h)))

    def submit(self, srcs_dev, dst_dev, stream=None):
        ptrs = (ctypes.c_void_p * len(srcs_dev))(*[_dev_ptr(s).value for s in srcs_dev])
        _check(lib().dfx_concat_submit(self._h, ptrs, _dev_ptr(dst_dev), _stream_ptr(stream)))

    def submit_gathered(self, gathered_dev, offsets, dst_dev, stream=None):
        offs = (ctypes.c_uint64 * len(offsets))(*offsets)
        _check(lib().dfx_concat_submit_gathered(self._h, _dev_ptr(gathered_dev), offs,
                                                _dev_ptr(dst_dev), _stream_ptr(stream)))

    def submit_host(self, srcs_np):
        srcs = [np.ascontiguousarray(s, dtype=self.np_dtype) for s in srcs_np]
        ptrs = (ctypes.c_void_p * len(srcs))(*[s.ctypes.data for s in srcs])
        dst = np.empty(self.dst_shape, dtype=self.np_dtype)
        _check(lib().dfx_concat_submit_host(self._h, ptrs, _p(dst)))
        return dst

    def close(self):
        if self._h:
            lib().dfx_concat_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
