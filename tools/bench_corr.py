"""timing of the correlation volume at the flow net's five level shapes (KITTI size), register-tiled kernel vs the first one
(DFVO_CORR_RT=0); torch.cuda events around 20 launches on the default stream"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
importlib.import_module("df-vo_amd")
capi = importlib.import_module("df-vo_amd.capi")
lib = capi.lib()
ptr = lambda t: capi.C.c_void_p(t.data_ptr()) if hasattr(capi, "C") else __import__("ctypes").c_void_p(t.data_ptr())
import ctypes
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
tot = 0.0
for name, C, stride, h, w in (("L6", 192, 1, 6, 19), ("L5", 128, 1, 11, 38), ("L4", 96, 1, 22, 76), ("L3", 64, 1, 44, 152), ("L2", 64, 2, 88, 304)):
    a = torch.randn(2, h, w, C, device="cuda")
    b = torch.randn(2, h, w, C, device="cuda")
    ho, wo = -(-h // stride), -(-w // stride)
    out = torch.zeros(2, ho, wo, 52, device="cuda")
    for _ in range(3):
        capi.check(lib.dfvo_correlation(ptr(a), ptr(b), 2, h, w, C, stride, 0.1, ptr(out), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        capi.check(lib.dfvo_correlation(ptr(a), ptr(b), 2, h, w, C, stride, 0.1, ptr(out), None))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print("%s C=%3d %3dx%3d stride %d: %7.1f us   checksum %.6f" % (name, C, h, w, stride, us, float(out.double().sum())))
print("RT=%s sum %.1f us" % (os.environ.get("DFVO_CORR_RT", "1"), tot))
