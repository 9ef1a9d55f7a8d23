"""per-layer timing of the f16x3 window layers at the KITTI pyramid sizes (level 2: 176x608, level 3: 88x304) for the
precision mode in PREC (f16x3 default | f16 | fp32); prints HIP-event durations and a CRC of the output (skeletons that
accumulate in the same order print equal CRCs)."""
import importlib
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
importlib.import_module("df-vo_amd")
capi = importlib.import_module("df-vo_amd.capi")
from test_ops_gpu import run_conv  # noqa: E402

lib = capi.lib()
LAYERS = [("L2 128->128", 2, 176, 608, 128, 0, 128), ("L2 64+66->128", 2, 176, 608, 64, 66, 128), ("L2 49->128", 2, 176, 608, 49, 0, 128),
          ("L2 128->64", 2, 176, 608, 128, 0, 64), ("L2 64->64", 2, 176, 608, 64, 0, 64), ("L2 64->32", 2, 176, 608, 64, 0, 32),
          ("L2 32->32", 2, 176, 608, 32, 0, 32), ("L3 128->128", 2, 88, 304, 128, 0, 128), ("L3 128->64", 2, 88, 304, 128, 0, 64),
          ("L3 64->32", 2, 88, 304, 64, 0, 32), ("c5 128->128", 2, 640, 960, 128, 0, 128),
          # (not in the KITTI sum) small maps: pyramid level 4 and the depth decoder's middle layers
          ("s L4 128->128", 2, 44, 152, 128, 0, 128), ("s L4 96+98->128", 2, 44, 152, 96, 98, 128), ("s L4 49->128", 2, 44, 152, 49, 0, 128),
          ("s L4 128->64", 2, 44, 152, 128, 0, 64), ("s L4 64->64", 2, 44, 152, 64, 0, 64), ("s L4 64->32", 2, 44, 152, 64, 0, 32),
          ("s D 128->64 48x160", 1, 48, 160, 128, 0, 64), ("s D 64+64->64 48x160", 1, 48, 160, 64, 64, 64), ("s D 256->128 24x80", 1, 24, 80, 256, 0, 128)]
g = torch.Generator().manual_seed(1)
capi.check(lib.dfvo_set_conv_precision(os.environ.get("PREC", "f16x3").encode()))
tot = 0.0
for name, n, h, w, c0, c1, cout in LAYERS:
    x0 = torch.randn(n, c0, h, w, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    # OPERANDS=zero_x / zero_all / const (timing only): the same instruction stream on operands that toggle fewer wires --
    # what the launch time owes to the power limit rather than to its schedule
    mode = os.environ.get("OPERANDS", "random")
    if mode in ("zero_x", "zero_all"):
        x0 = torch.zeros_like(x0)
        x1 = torch.zeros_like(x1) if x1 is not None else None
    if mode == "zero_all":
        wt = torch.zeros_like(wt)
    if mode == "const":
        x0 = torch.full_like(x0, 0.5)
        x1 = torch.full_like(x1, 0.5) if x1 is not None else None
        wt = torch.full_like(wt, 0.25)
    b = torch.zeros(cout)
    gf = 2.0 * n * h * w * 9 * (c0 + c1) * cout / 1e9
    best = 1e9
    for rep in range(3):
        capi.check(lib.dfvo_conv_profile_begin())
        out = run_conv(capi, x0, wt, b, 1, (1, 1), 0, 1, 0.1, x1=x1)
        ms, fl, ln = np.zeros(24), np.zeros(24), np.zeros(24, np.int32)
        capi.check(lib.dfvo_conv_profile_end(capi.as_ptr(ms), capi.as_ptr(fl), capi.as_ptr(ln)))
        best = min(best, ms.sum())
    if not name.startswith("c5") and not name.startswith("s "):
        tot += best
    print("%-16s %6.1f GF %8.1f us %6.1f TF/s-eq  crc %08x" % (name, gf, best * 1e3, gf / best, zlib.crc32(out.numpy().tobytes()) & 0xffffffff), flush=True)
print("PREC=%s DFVO_WIN=%s operands %s: sum of the KITTI layers %.1f us" % (os.environ.get("PREC", "f16x3"), os.environ.get("DFVO_WIN", "-"), os.environ.get("OPERANDS", "random"), tot * 1e3))
