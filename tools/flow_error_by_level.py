"""Where the flow net's distance to the exact function enters: per pyramid level, |device - float64 anchor| beside
|torch-CPU fp32 - float64 anchor| (oracle/nets_torch.py, dtype=float64: same fp32 inputs / weights / grid constants, every
operation in double).  Test-side tool (imports the oracle); the product never does.

    DFVO_CONV_PRECISION=fp32|f16x3 python tools/flow_error_by_level.py [--world tunnel|random]
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nets_torch as O  # noqa: E402
import test_nets_gpu as T  # noqa: E402


def stats(x, exact):
    e = np.abs(np.asarray(x, np.float64) - np.asarray(exact, np.float64))
    return "max %.2e p99 %.2e median %.2e" % (e.max(), np.quantile(e, 0.99), np.median(e))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", default="tunnel")
    a = ap.parse_args()
    capi = importlib.import_module("df-vo_amd.capi")
    lib = capi.lib()
    if a.world == "tunnel":
        h, w = 256, 640
        syn = importlib.import_module("df-vo_amd.synthetic")
        seq = syn.coded_tunnel_sequence(h, w, 3, mode="mux", step=1.0, seed=21)
        sd = syn.crafted_liteflownet_state_dict(h, w, "mux")
        ref_img, cur_img = seq["frames"][1], seq["frames"][2]
    else:
        from synth import image_pair
        h, w = 192, 640
        sd = O.liteflownet_state_dict(4869)
        ref_img, cur_img = image_pair(h, w, seed=1001 + h)
    net, nh, nw = T.make_flownet(capi, h, w, sd)
    fwd, bwd, diff = np.zeros((2, h, w), np.float32), np.zeros((2, h, w), np.float32), np.zeros((h, w), np.float32)
    capi.check(lib.dfvo_flownet_forward_host(net, capi.as_ptr(ref_img), capi.as_ptr(cur_img), capi.as_ptr(fwd), capi.as_ptr(bwd),
                                             capi.as_ptr(diff)))
    O._grid_cache.clear()
    f32, b32, d32, raw32 = O.flow_inference(sd, ref_img, cur_img, return_levels=True)
    f64, b64, d64, raw64 = O.flow_inference(sd, ref_img, cur_img, return_levels=True, dtype=torch.float64)
    print("world %s, %dx%d (net %dx%d), conv precision %s" % (a.world, h, w, nh, nw, os.environ.get("DFVO_CONV_PRECISION", "fp32")))
    for lvl in (6, 5, 4, 3, 2):
        lh, lw = nh >> (lvl - 1), nw >> (lvl - 1)
        buf = np.zeros((2, lh, lw, 2), np.float32)
        capi.check(lib.dfvo_flownet_get_level_flow(net, lvl, capi.as_ptr(buf), None, None))
        dev = np.transpose(buf, (0, 3, 1, 2))
        ex = raw64[lvl].numpy()
        for smp in (0, 1):
            print("level %d raw flow, sample %d (max |flow| %.2f): device-exact %s | oracle32-exact %s"
                  % (lvl, smp, np.abs(ex[smp]).max(), stats(dev[smp], ex[smp]), stats(raw32[lvl].numpy()[smp], ex[smp])))
            if lvl == 2:  # signed structure of the error: per channel mean / std, and the mean over eight column bands
                for c in (0, 1):
                    e = dev[smp, c].astype(np.float64) - ex[smp, c]
                    eo = raw32[lvl].numpy()[smp, c].astype(np.float64) - ex[smp, c]
                    bands = np.array_split(np.arange(e.shape[1]), 8)
                    print("      ch %d: device mean %+.2e std %.2e | oracle32 mean %+.2e std %.2e | exact flow mean %+.3f | device error by column band: %s"
                          % (c, e.mean(), e.std(), eo.mean(), eo.std(), ex[smp, c].mean(), " ".join("%+.1e" % e[:, b].mean() for b in bands)))
    for name, d, o, e in (("fwd", fwd, f32, f64), ("bwd", bwd, b32, b64), ("diff", diff, d32[..., 0], d64[..., 0])):
        print("%s (max %.2f): device-exact %s | oracle32-exact %s" % (name, np.abs(e).max(), stats(d, e), stats(o, e)))
    lib.dfvo_flownet_destroy(net)


if __name__ == "__main__":
    main()
