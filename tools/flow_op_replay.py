"""Per-operator accuracy of the flow net's level-2 operators on REAL intermediate data (test-side tool, imports the oracle).

The oracle runs once in float64 (oracle/nets_torch.py, dtype=float64) with its primitives hooked; every recorded call of
the chosen pyramid level is then replayed in isolation on the float32-rounded inputs: (a) torch CPU fp32, (b) the device
operator through the C ABI (dfvo_conv2d / dfvo_backward_warp / dfvo_correlation / dfvo_deconv_dw4x4s2), each compared with
the float64 result of the same rounded inputs.  Signed mean, median and max |error| per batch sample: a biased operator
shows up as a signed mean of the size of its median.

    DFVO_CONV_PRECISION=fp32|f16x3 python tools/flow_op_replay.py [--level 2] [--world tunnel|random]
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nets_torch as O  # noqa: E402
import test_ops_gpu as TO  # noqa: E402
from util import nhwc_dev, nchw_host, ptr  # noqa: E402

REC = []
CTX = {"stage": "features", "level": 0, "on": True}


def hooked(name, fn):
    def w(*a, **k):
        if not CTX["on"]:
            return fn(*a, **k)
        CTX["on"] = False
        try:
            out = fn(*a, **k)
        finally:
            CTX["on"] = True
        REC.append((name, CTX["stage"], CTX["level"], a, k, out))
        return out
    return w


class FProxy:
    def __init__(self):
        self.conv2d = hooked("conv2d", F.conv2d)
        self.conv_transpose2d = hooked("deconv", F.conv_transpose2d)

    def __getattr__(self, n):
        return getattr(F, n)


def staged(name, fn):
    def w(sd, lvl, *a, **k):
        CTX["stage"], CTX["level"] = name, lvl
        return fn(sd, lvl, *a, **k)
    return w


def line(tag, got, exact):
    e = got.double() - exact
    s = []
    for n in range(e.shape[0]):
        a = e[n].abs()
        s.append("s%d mean %+.2e med %.2e max %.2e" % (n, e[n].mean().item(), a.median().item(), a.max().item()))
    return "%-8s %s" % (tag, " | ".join(s))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--world", default="tunnel")
    ap.add_argument("--dry", action="store_true", help="no device: the torch fp32 column only")
    a = ap.parse_args()
    capi = importlib.import_module("df-vo_amd.capi")
    lib = None if a.dry else capi.lib()
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
    O.F = FProxy()
    O.backward_warp = hooked("warp", O.backward_warp)
    O.correlation = hooked("corr", O.correlation)
    O.matching, O.subpixel, O.regularization = staged("M", O.matching), staged("S", O.subpixel), staged("R", O.regularization)
    O._grid_cache.clear()
    O.flow_inference(sd, ref_img, cur_img, dtype=torch.float64)
    CTX["on"] = False
    print("world %s, level %d, conv precision %s; errors vs float64 on the SAME float32-rounded inputs" % (a.world, a.level, os.environ.get("DFVO_CONV_PRECISION", "fp32")))
    for name, stage, lvl, args, kw, out in REC:
        if lvl != a.level:
            continue
        if name == "conv2d":
            x, wt = args[0].float(), args[1].float()
            b = args[2] if len(args) > 2 else kw.get("bias")
            b = b.float() if b is not None else None
            stride = kw.get("stride", args[3] if len(args) > 3 else 1)
            pad = kw.get("padding", args[4] if len(args) > 4 else 0)
            pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
            exact = F.conv2d(x.double(), wt.double(), b.double() if b is not None else None, stride=stride, padding=pad)
            t32 = F.conv2d(x, wt, b, stride=stride, padding=pad)
            dev = t32 if a.dry else TO.run_conv(capi, x, wt, b if b is not None else torch.zeros(wt.shape[0]), stride, pad)
            desc = "%s conv %dx%d s%d %d->%d |out| %.2e" % (stage, wt.shape[2], wt.shape[3], stride, wt.shape[1], wt.shape[0], exact.abs().max().item())
        elif name == "deconv":
            x, wt = args[0].float(), args[1].float()
            exact = F.conv_transpose2d(x.double(), wt.double(), None, stride=2, padding=1, groups=x.shape[1])
            t32 = F.conv_transpose2d(x, wt, None, stride=2, padding=1, groups=x.shape[1])
            n, c, hh, ww = x.shape
            cs = (c + 3) // 4 * 4
            if a.dry:
                print("%s deconv C=%d" % (stage, c)); print("   " + line("torch32", t32, exact)); continue
            dx = nhwc_dev(x, cs)
            dst = torch.zeros(n, 2 * hh, 2 * ww, cs, device="cuda")
            wn = np.ascontiguousarray(wt.numpy())
            capi.check(lib.dfvo_deconv_dw4x4s2(ptr(dx), n, hh, ww, c, cs, capi.as_ptr(wn), ptr(dst), None))
            torch.cuda.synchronize()
            dev = nchw_host(dst, c)
            desc = "%s deconv 4x4 s2 C=%d |out| %.2e" % (stage, c, exact.abs().max().item())
        elif name == "warp":
            src, fl = args[0].float(), args[1].float()
            exact = O.backward_warp(src.double(), fl.double())
            t32 = O.backward_warp(src, fl)
            n, c, hh, ww = src.shape
            if a.dry:
                print("%s warp C=%d" % (stage, c)); print("   " + line("torch32", t32, exact)); continue
            lx = torch.linspace(-1.0, 1.0, ww).numpy()
            ly = torch.linspace(-1.0, 1.0, hh).numpy()
            dsrc = nhwc_dev(src)
            dflow = fl.permute(0, 2, 3, 1).contiguous().cuda()
            dst = torch.zeros(n, hh, ww, dsrc.shape[3], device="cuda")
            capi.check(lib.dfvo_backward_warp(ptr(dsrc), ptr(dflow), 1.0, n, hh, ww, dsrc.shape[3], capi.as_ptr(lx), capi.as_ptr(ly), ptr(dst), None))
            torch.cuda.synchronize()
            dev = nchw_host(dst, c)
            desc = "%s warp C=%d |flow| %.2f |out| %.2e" % (stage, c, fl.abs().max().item(), exact.abs().max().item())
        elif name == "corr":
            f1, f2, stride = args[0].float(), args[1].float(), args[2]
            exact = O.correlation(f1.double(), f2.double(), stride)
            t32 = O.correlation(f1, f2, stride)
            n, c, hh, ww = f1.shape
            if a.dry:
                print("%s corr C=%d" % (stage, c)); print("   " + line("torch32", t32, exact)); continue
            ho, wo = -(-hh // stride), -(-ww // stride)
            out_d = torch.zeros(n, ho, wo, 52, device="cuda")
            d1, d2 = nhwc_dev(f1), nhwc_dev(f2)  # (named: a temporary's block would be handed to the next allocation)
            capi.check(lib.dfvo_correlation(ptr(d1), ptr(d2), n, hh, ww, c, stride, 1.0, ptr(out_d), None))
            torch.cuda.synchronize()
            dev = nchw_host(out_d, 49)
            desc = "%s correlation C=%d s%d |out| %.2e" % (stage, c, stride, exact.abs().max().item())
        else:
            continue
        print(desc)
        print("   " + line("torch32", t32, exact))
        print("   " + line("device", dev, exact))


if __name__ == "__main__":
    main()
