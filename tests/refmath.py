"""Independent formulation of the fused conv, used only to pin the C oracle.

Works from plain OIHW weights (so it also checks the OIhw4i16o4i reorder),
computes exact accumulators with torch float64 conv2d (exact: |acc| < 2**53) and
performs the requantisation with numpy float32 element-wise ops (one rounding per
op, no FMA), following SURVEY.md 8(a) "Intended semantics".
"""
import numpy as np
import torch
import torch.nn.functional as F

from cases import F32, S32, S8, U8, UNDEF


def _acc_conv(src_nhwc, w_oihw, stride, pad):
    x = torch.from_numpy(src_nhwc.astype(np.float64)).permute(0, 3, 1, 2)
    w = torch.from_numpy(w_oihw.astype(np.float64))
    y = F.conv2d(x, w, stride=stride, padding=pad)          # zero padding
    return y.permute(0, 2, 3, 1).contiguous().numpy().astype(np.int64)


def _cvt_x86(f, rm):
    """vcvtps2dq: RNE or floor; NaN / out of range -> 0x80000000."""
    r = np.floor(f) if rm == 1 else np.rint(f)
    bad = ~((f >= np.float32(-2147483648.0)) & (f < np.float32(2147483648.0)))
    out = np.where(bad, 0.0, r).astype(np.int64)
    out[bad] = -2147483648
    return out


def _requant(acc, bias, scales, relu):
    f = acc.astype(np.float32)                               # int -> f32, RNE
    if bias is not None:
        f = f + bias.astype(np.float32)
    f = f * scales.astype(np.float32)
    if relu:
        f = np.where(np.float32(0) > f, np.float32(0), f)    # vmaxps(zero, f)
    return f.astype(np.float32)


def _store(f, dt, rm):
    if dt == F32:
        return f
    v = _cvt_x86(f, rm)
    if dt == S32:
        return v.astype(np.int32)
    if dt == S8:
        return np.clip(v, -128, 127).astype(np.int8)
    return np.where((v < 0) | (v > 255), 255, v).astype(np.uint8)   # unsigned saturate of the bits


def conv_ref(case, data):
    acc0 = _acc_conv(data["src"], data["w0"], case.stride, case.pad)
    fused = case.oc1x1 > 0
    relu0 = fused or case.relu0 or case.dst_dt == U8
    f0 = _requant(acc0, data["bia0"], data["scales0"], relu0)
    if not fused:
        return _store(f0, case.dst_dt, case.rm0)
    mid = _store(f0, U8, case.rm0).astype(np.int64)
    w1 = data["w1"].reshape(case.oc1x1, case.oc).astype(np.int64)
    acc1 = mid @ w1.T
    relu1 = case.relu1 or case.dst_dt == U8
    f1 = _requant(acc1, data["bia1"], data["scales1"], relu1)
    return _store(f1, case.dst_dt, case.rm1)


def maxpool(src, kernel, stride, pad, out_hw):
    """Independent formulation of NHWC max pooling (padding takes no part): pad with the dtype's lowest
    value, then maximum over shifted strided views -- no loop over output positions."""
    bs, ih, iw, c = src.shape
    oh, ow = out_hw
    low = -np.inf if src.dtype == np.float32 else np.iinfo(src.dtype).min
    need_h = (oh - 1) * stride[0] + kernel[0]
    need_w = (ow - 1) * stride[1] + kernel[1]
    buf = np.full((bs, max(need_h, pad[0] + ih), max(need_w, pad[1] + iw), c), low, dtype=src.dtype)
    buf[:, pad[0]:pad[0] + ih, pad[1]:pad[1] + iw, :] = src
    out = None
    for ky in range(kernel[0]):
        for kx in range(kernel[1]):
            v = buf[:, ky:ky + (oh - 1) * stride[0] + 1:stride[0], kx:kx + (ow - 1) * stride[1] + 1:stride[1], :]
            out = v.copy() if out is None else np.maximum(out, v)
    return out


def avgpool(src, kernel, stride, pad, out_hw, include_padding):
    """Independent formulation of NHWC average pooling: zero-padded buffer and a ones mask, both summed over
    shifted strided views; integer types divide in f32 and round half to even.  (f32: the window-order sum
    of the oracle equals this shifted-view order: ky outer, kx inner.)"""
    bs, ih, iw, c = src.shape
    oh, ow = out_hw
    need_h = (oh - 1) * stride[0] + kernel[0]
    need_w = (ow - 1) * stride[1] + kernel[1]
    is_f = src.dtype == np.float32
    buf = np.zeros((bs, max(need_h, pad[0] + ih), max(need_w, pad[1] + iw), c), dtype=np.float32 if is_f else np.int64)
    msk = np.zeros(buf.shape[1:3], dtype=np.int64)
    buf[:, pad[0]:pad[0] + ih, pad[1]:pad[1] + iw, :] = src
    msk[pad[0]:pad[0] + ih, pad[1]:pad[1] + iw] = 1
    tot = np.zeros((bs, oh, ow, c), dtype=buf.dtype)
    cnt = np.zeros((oh, ow), dtype=np.int64)
    for ky in range(kernel[0]):
        for kx in range(kernel[1]):
            sl = (slice(ky, ky + (oh - 1) * stride[0] + 1, stride[0]), slice(kx, kx + (ow - 1) * stride[1] + 1, stride[1]))
            m = msk[sl]
            if is_f:  # skip padded positions entirely (adding +0.0 could turn a -0.0 sum into +0.0)
                tot = np.where(m[None, :, :, None] == 1, (tot + buf[(slice(None),) + sl]).astype(np.float32), tot)
            else:
                tot = tot + buf[(slice(None),) + sl]
            cnt = cnt + m
    div = np.full((oh, ow), kernel[0] * kernel[1], dtype=np.int64) if include_padding else cnt
    if is_f:
        return (tot / div[None, :, :, None].astype(np.float32)).astype(np.float32)
    q = np.rint(tot.astype(np.float32) / div[None, :, :, None].astype(np.float32))
    info = np.iinfo(src.dtype)
    return np.clip(q, info.min, info.max).astype(src.dtype)


def eltwise_sum(srcs, post_relu=False):
    """Independent formulation: exact integer sum in int64 (f32: left to right in f32), clip, relu."""
    if srcs[0].dtype == np.float32:
        acc = srcs[0].copy()
        for s in srcs[1:]:
            acc = (acc + s).astype(np.float32)
        return np.where(acc < 0, np.float32(0), acc).astype(np.float32) if post_relu else acc
    acc = np.zeros(srcs[0].shape, dtype=np.int64)
    for s in srcs:
        acc += s.astype(np.int64)
    if post_relu:
        acc = np.maximum(acc, 0)
    info = np.iinfo(srcs[0].dtype)
    return np.clip(acc, info.min, info.max).astype(srcs[0].dtype)
