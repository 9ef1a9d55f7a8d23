"""Times the two CNN executors on synthetic KITTI-sized frames (device-resident inputs)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.dfvo_amd()
import importlib  # noqa: E402

capi = importlib.import_module("df-vo_amd.capi")
from oracle import nets_torch as O  # noqa: E402
from synth import image_pair  # noqa: E402
from test_nets_gpu import make_flownet  # noqa: E402


def main():
    h, w = int(os.environ.get("H", 376)), int(os.environ.get("W", 1241))
    steps = int(os.environ.get("STEPS", 20))
    lib = capi.lib()
    capi.require_gpu()
    sd = O.liteflownet_state_dict(4869)
    ref, cur = image_pair(h, w, seed=1)
    net, nh, nw = make_flownet(capi, h, w, sd, graph=int(os.environ.get("GRAPH", 1)))
    dref = torch.from_numpy(ref).cuda()
    dcur = torch.from_numpy(cur).cuda()
    fwd = torch.zeros(2, h, w, device="cuda")
    bwd = torch.zeros(2, h, w, device="cuda")
    diff = torch.zeros(h, w, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        capi.check(lib.dfvo_flownet_forward(net, p(dref), p(dcur), p(fwd), p(bwd), p(diff)))
    capi.check(lib.dfvo_flownet_sync(net))
    t0 = time.time()
    for _ in range(steps):
        capi.check(lib.dfvo_flownet_forward(net, p(dref), p(dcur), p(fwd), p(bwd), p(diff)))
    capi.check(lib.dfvo_flownet_sync(net))
    dt = (time.time() - t0) / steps
    fl = lib.dfvo_flownet_last_flops(net)
    print("flownet %dx%d (net %dx%d): %.3f ms/pair, %.1f GFLOP -> %.1f TFLOP/s" % (h, w, nh, nw, dt * 1e3, fl / 1e9,
                                                                                   fl / dt / 1e12))
    # depth
    sdd = O.monodepth2_state_dict(4869)
    dn = C.c_void_p()
    capi.check(lib.dfvo_depthnet_create(192, 640, 0.1, 100.0, 5.4, None, C.byref(dn)))
    capi.set_params(lib.dfvo_depthnet_set_param, dn, {k: v.numpy() for k, v in sdd.items()})
    capi.check(lib.dfvo_depthnet_finalize(dn))
    dimg = torch.from_numpy(image_pair(192, 640, seed=2)[0]).cuda()
    dd = torch.zeros(192, 640, device="cuda")
    for _ in range(3):
        capi.check(lib.dfvo_depthnet_forward(dn, p(dimg), p(dd)))
    capi.check(lib.dfvo_depthnet_sync(dn))
    t0 = time.time()
    for _ in range(steps):
        capi.check(lib.dfvo_depthnet_forward(dn, p(dimg), p(dd)))
    capi.check(lib.dfvo_depthnet_sync(dn))
    dt2 = (time.time() - t0) / steps
    fl2 = lib.dfvo_depthnet_last_flops(dn)
    print("depthnet 192x640: %.3f ms, %.1f GFLOP -> %.1f TFLOP/s" % (dt2 * 1e3, fl2 / 1e9, fl2 / dt2 / 1e12))


if __name__ == "__main__":
    main()
