"""why does a pair track / not track in the f16 net mode?  python tools/f16_mode_debug.py [h w mode]: one pair of the coded
tunnel world through the pipeline in f16x3 and f16, consistency-map statistics and the tracker's diagnostics"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("df-vo_amd")
capi = importlib.import_module("df-vo_amd.capi")
pmod = importlib.import_module("df-vo_amd.pipeline")
smod = importlib.import_module("df-vo_amd.sequence")
syn = importlib.import_module("df-vo_amd.synthetic")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 640)
mode = sys.argv[3] if len(sys.argv) > 3 else "mux"
poses = None if mode == "mux" else syn.tunnel_poses_lateral(4, 0.4)
seq = syn.coded_tunnel_sequence(h, w, 4, mode=mode, step=1.0, seed=21, poses=poses)
for prec in ("f16x3", "f16"):
    capi.check(capi.lib().dfvo_set_conv_precision(prec.encode()))
    pipe = pmod.TrackingPipeline(h, w, 192, 640, seq["K"], syn.crafted_liteflownet_state_dict(h, w, mode),
                                 syn.crafted_monodepth2_state_dict(), seed=4869)
    fr = smod.frames_to_device(seq["frames"])
    pipe.set_ref_image(fr[0])
    for k in range(2):
        pipe.enqueue_nets(k, fr[k], fr[k + 1])
        out = pipe.track(k)
        fwd, bwd, diff, raw, dep = pipe.get_outputs(k)
        d = diff.reshape(-1)
        print("%s %dx%d %s pair %d: diff median %.4f p90 %.4f p99 %.4f max %.3f | count(diff < 0.1) = %d of %d | |fwd| max %.2f | status %d "
              "good_kp %d n_kp %d inliers %d scale %.4f | depth median %.2f" % (
                  prec, h, w, mode, k, np.median(d), np.quantile(d, 0.9), np.quantile(d, 0.99), d.max(), int((d < 0.1).sum()), d.size,
                  np.abs(fwd).max(), out.status, out.good_kp_found, out.n_kp, out.best_inlier_cnt, out.scale, np.median(dep[dep > 0]) if (dep > 0).any() else -1))
    pipe.close()
capi.check(capi.lib().dfvo_set_conv_precision(b"fp32"))
