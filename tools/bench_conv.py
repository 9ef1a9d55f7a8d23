"""Micro-benchmark of one convolution shape through dfvo_conv2d (weights packed per call, outside the timed events).
env: N H W C0 C1 COUT K (or KH KW) STRIDE ITERS, DFVO_CONV_PRECISION.  Prints us per launch and TFLOP/s from the library's own
per-launch HIP events.  Under `rocprofv3 --pmc ...` this is the single-layer target of the counter passes."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("df-vo_amd.capi")


def main():
    e = lambda k, d: int(os.environ.get(k, d))
    N, H, W, C0, C1, COUT, K, S, IT = e("N", 2), e("H", 192), e("W", 624), e("C0", 128), e("C1", 0), e("COUT", 128), e("K", 3), e("STRIDE", 1), e("ITERS", 20)
    KH, KW = e("KH", K), e("KW", K)
    lib = capi.lib()
    capi.require_gpu()
    cs0 = (C0 + 3) // 4 * 4
    cs1 = (C1 + 3) // 4 * 4 if C1 else 0
    x0 = torch.randn(N, H, W, cs0, device="cuda")
    x1 = torch.randn(N, H, W, cs1, device="cuda") if C1 else None
    ph, pw = (KH - 1) // 2, (KW - 1) // 2
    Ho = (H + 2 * ph - KH) // S + 1
    Wo = (W + 2 * pw - KW) // S + 1
    dcs = (COUT + 3) // 4 * 4
    dst = torch.zeros(N, Ho, Wo, dcs, device="cuda")
    w = (np.random.randn(COUT, C0 + C1, KH, KW) * 0.05).astype(np.float32)
    b = np.zeros(COUT, np.float32)
    desc = capi.ConvDesc(N=N, H=H, W=W, kh=KH, kw=KW, stride=S, pad_h=ph, pad_w=pw, pad_mode=0, c0=C0, cs0=cs0, co0=0, up0=0,
                         c1=C1, cs1=cs1, co1=0, cout=COUT, act=1, act_param=0.1, res_cs=0, res_co=0, dst_cs=dcs, dst_co=0)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    for _ in range(2):
        capi.check(lib.dfvo_conv2d(C.byref(desc), p(x0), p(x1), capi.as_ptr(w), capi.as_ptr(b), None, p(dst), None))
    ms = np.zeros(24)
    fl = np.zeros(24)
    ln = np.zeros(24, np.int32)
    capi.check(lib.dfvo_conv_profile_begin())
    for _ in range(IT):
        capi.check(lib.dfvo_conv2d(C.byref(desc), p(x0), p(x1), capi.as_ptr(w), capi.as_ptr(b), None, p(dst), None))
    capi.check(lib.dfvo_conv_profile_end(capi.as_ptr(ms), capi.as_ptr(fl), capi.as_ptr(ln)))
    flops = 2.0 * N * Ho * Wo * COUT * (C0 + C1) * KH * KW
    i = int(np.argmax(ms))
    print("conv N%d %dx%d c%d+%d->%d k%dx%d s%d: cfg %d, %.1f us/launch, %.1f TFLOP/s" % (
        N, H, W, C0, C1, COUT, KH, KW, S, i, ms[i] * 1e3 / ln[i], flops * ln[i] / (ms[i] * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
