"""Accuracy and speed of the opt-in split-precision (bf16x3) conv mode against the exact fp32 path, on the flow net
(random KITTI-sized pair, random weights) and on the hottest layer shape.  Run twice:
    python tools/split_precision_check.py                         (fp32)
    DFVO_CONV_PRECISION=bf16x3 python tools/split_precision_check.py
The first run stores its outputs in /tmp/flow_fp32.npz, the second compares against them."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("df-vo_amd.capi")
syn = importlib.import_module("df-vo_amd.synthetic")
mode = os.environ.get("DFVO_CONV_PRECISION", "fp32")
h, w = 376, 1241
lf = importlib.import_module("df-vo_amd.libs.deep_models.flow.lite_flow_net.lite_flow").LiteFlow(h, w)
lf.initialize_network_model(syn.liteflownet_state_dict(4869), False)
ref, cur = syn.image_pair(h, w, seed=77)
fwd, bwd, diff = lf.inference_flow_u8(ref, cur)
t0 = time.time()
for _ in range(10):
    lf.inference_flow_u8(ref, cur)
dt = (time.time() - t0) / 10
print("mode %s: flow net %.2f ms per pair incl. host copies; |fwd| max %.2f" % (mode, dt * 1e3, np.abs(fwd).max()))
path = "/tmp/flow_fp32.npz"
if mode == "fp32":
    np.savez(path, fwd=fwd, bwd=bwd, diff=diff)
elif os.path.exists(path):
    g = np.load(path)
    for k, v in (("fwd", fwd), ("bwd", bwd), ("diff", diff)):
        e = np.abs(v - g[k])
        print("  %s vs fp32: max |err| %.3e px, mean %.3e, p99.9 %.3e" % (k, e.max(), e.mean(), np.quantile(e, 0.999)))
