"""per-layer timing of the 3x3 window layers of LiteFlowNet's level 2 / 3 at KITTI size: exact fp32 MFMA kernel vs the
f16x3 split kernel (conv_win_f16s_kernel); HIP-event durations from the library's profiling hooks."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
importlib.import_module("df-vo_amd")
capi = importlib.import_module("df-vo_amd.capi")
from test_ops_gpu import run_conv  # noqa: E402

lib = capi.lib()
LAYERS = [("L2 128->128", 2, 192, 624, 128, 0, 128), ("L2 64+66->128", 2, 192, 624, 64, 66, 128), ("L2 3+128->128", 2, 192, 624, 3, 128, 128),
          ("L2 49->128", 2, 192, 624, 49, 0, 128), ("L2 128->64", 2, 192, 624, 128, 0, 64), ("L2 64->64", 2, 192, 624, 64, 0, 64),
          ("L2 64->32", 2, 192, 624, 64, 0, 32), ("L2 32->32", 2, 192, 624, 32, 0, 32), ("L3 128->128", 2, 96, 312, 128, 0, 128),
          ("L3 128->64", 2, 96, 312, 128, 0, 64), ("L4 128->128", 2, 48, 156, 128, 0, 128)]
if os.environ.get("ONLY"):
    LAYERS = [l for l in LAYERS if l[0] == os.environ["ONLY"]]
g = torch.Generator().manual_seed(1)
for name, n, h, w, c0, c1, cout in LAYERS:
    x0 = torch.randn(n, c0, h, w, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    b = torch.zeros(cout)
    gf = 2.0 * n * h * w * 9 * (c0 + c1) * cout / 1e9
    line = "%-16s %6.1f GF" % (name, gf)
    outs = {}
    for mode in (b"fp32", b"f16x3"):
        capi.check(lib.dfvo_set_conv_precision(mode))
        best = 1e9
        for rep in range(3):
            capi.check(lib.dfvo_conv_profile_begin())
            outs[mode] = run_conv(capi, x0, wt, b, 1, (1, 1), 0, 1, 0.1, x1=x1)
            ms, fl, ln = np.zeros(24), np.zeros(24), np.zeros(24, np.int32)
            capi.check(lib.dfvo_conv_profile_end(capi.as_ptr(ms), capi.as_ptr(fl), capi.as_ptr(ln)))
            best = min(best, ms.sum())
        line += " | %s %7.1f us %6.1f TF/s (cfg %d)" % (mode.decode(), best * 1e3, gf / best, int(np.argmax(ms)))
    capi.check(lib.dfvo_set_conv_precision(b"fp32"))
    d = (outs[b"fp32"] - outs[b"f16x3"]).abs().max().item() / outs[b"fp32"].abs().max().item()
    print(line + " | rel diff %.2e" % d, flush=True)
