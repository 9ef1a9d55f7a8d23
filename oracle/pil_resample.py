"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of Pillow's 8-bit two-pass resampling with the LANCZOS filter, the host-side resize the reference
applies to the depth-net input (/root/reference/libs/deep_models/deep_models.py:195-199, `img.resize(..., pil.LANCZOS)`).
Pillow is a third-party dependency of the reference (PIL 6.0); the algorithm restated here is
its src/libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc (22-bit fixed-point coefficients),
ImagingResampleHorizontal_8bpc then ImagingResampleVertical_8bpc, each pass rounding to uint8 through clip8.

Pinned: tests/test_oracle_lanczos.py compares this restatement bit-for-bit with the Pillow installed in the image
(`Image.resize(size, Image.LANCZOS)`) on random and structured images, and with tests/golden/lanczos_*.npz.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size, out_size, support_base=3.0, filt=_lanczos):
    """Resample.c precompute_coeffs with the full box (in0 = 0, in1 = in_size): bounds [out,2], integer coeffs [out,ksize]"""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = support_base * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = filt((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(xmax):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, xmax)
    # normalize_coeffs_8bpc: (int) truncation toward zero after the +-0.5
    ki = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int32)
    return bounds, ki


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass(img, bounds, ki):
    """resample axis 1 of img [A, B, C] -> [A, out, C] (int32 accumulation from 1 << (PRECISION_BITS - 1))"""
    out = np.empty((img.shape[0], bounds.shape[0], img.shape[2]), np.uint8)
    src = img.astype(np.int32)
    for xx in range(bounds.shape[0]):
        xmin, xmax = bounds[xx]
        acc = np.full((img.shape[0], img.shape[2]), 1 << (PRECISION_BITS - 1), np.int32)
        for x in range(xmax):
            acc = acc + src[:, xmin + x, :] * ki[xx, x]
        out[:, xx, :] = _clip8(acc)
    return out


def resize_lanczos_u8(img, out_w, out_h):
    """img uint8 [H, W, C] -> uint8 [out_h, out_w, C]: horizontal pass, then vertical pass (ImagingResample order)"""
    H, W = img.shape[:2]
    cur = img
    if out_w != W:
        b, k = precompute_coeffs(W, out_w)
        cur = _pass(cur, b, k)
    if out_h != H:
        b, k = precompute_coeffs(H, out_h)
        cur = _pass(cur.transpose(1, 0, 2), b, k).transpose(1, 0, 2)
    return np.ascontiguousarray(cur)
