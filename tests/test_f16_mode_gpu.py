"""GPU: the single-product "f16" net mode (dfvo_set_conv_precision("f16"): BASELINE.json config 5's "fp16 flow"), reported
SEPARATELY from the fp32-class modes as SURVEY.md section 7 (hard part 4) asks: "keep an fp32-accumulate / fp32-activation
parity mode for correctness gates and report fp16 separately with pose / RPE deltas".

What the mode computes is defined operator by operator (tests/test_ops_gpu.py::test_conv_f16_mode: the float64 convolution
of the f16-ROUNDED operands, to fp32 summation noise).  Here, on the whole nets and the whole tracking path:
  * the flow net's distance to the float64 anchor next to the oracle's (fp32 torch-CPU) own distance -- lines "F16-MODE ANCHOR",
  * the 130-frame coded tunnel tracked end to end: keypoint-set overlap with the fp32 oracle fixture, per-pair pose distance,
    t_rel / r_rel / ATE and their deltas to the oracle's -- lines "F16-MODE TRAJECTORY",
  * the solver stage stays bit-exact against the oracle chain on the device's own (f16-mode) arrays:
    tests/test_e2e_gpu.py::test_pipeline_config5_settings[f16].
Gates are the ones a lower-precision mode can honestly carry (set from the first measured run, margins stated); nothing
here feeds the headline number."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import kitti_eval as E
from oracle import nets_torch as O
from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, image_pair
from test_nets_gpu import _err_stats, _oracle_cache, _oracle_flow, make_flownet

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tunnel_traj.npz")


@pytest.mark.parametrize("world", ["random_weights_192x640", "coded_tunnel_256x640"])
def test_f16_flow_distance_to_the_exact_function(gpu, f16_mode, world):
    lib = gpu.lib()
    if world.startswith("random"):
        h, w = 192, 640
        sd = O.liteflownet_state_dict(4869)
        ref_img, cur_img = image_pair(h, w, seed=1001 + h)
    else:
        h, w = 256, 640
        syn = importlib.import_module("df-vo_amd.synthetic")
        seq = syn.coded_tunnel_sequence(h, w, 3, mode="mux", step=1.0, seed=21)
        sd = syn.crafted_liteflownet_state_dict(h, w, "mux")
        ref_img, cur_img = seq["frames"][1], seq["frames"][2]
    net, nh, nw = make_flownet(gpu, h, w, sd)
    fwd = np.zeros((2, h, w), np.float32)
    bwd = np.zeros((2, h, w), np.float32)
    diff = np.zeros((h, w), np.float32)
    gpu.check(lib.dfvo_flownet_forward_host(net, gpu.as_ptr(ref_img), gpu.as_ptr(cur_img), gpu.as_ptr(fwd), gpu.as_ptr(bwd),
                                            gpu.as_ptr(diff)))
    lib.dfvo_flownet_destroy(net)
    o32 = _oracle_flow(sd, ref_img, cur_img, ("anchor32", world))
    key = ("anchor64", world)
    if key not in _oracle_cache:
        O._grid_cache.clear()
        _oracle_cache[key] = O.flow_inference(sd, ref_img, cur_img, dtype=torch.float64)
    o64 = _oracle_cache[key]
    assert np.isfinite(fwd).all() and np.isfinite(bwd).all() and np.isfinite(diff).all()
    for name, dev, a32, a64 in (("fwd", fwd, o32[0], o64[0]), ("bwd", bwd, o32[1], o64[1]), ("diff", diff, o32[2][..., 0], o64[2][..., 0])):
        d, o = _err_stats(dev, a64), _err_stats(a32, a64)
        mag = float(np.abs(a64).max())
        print("F16-MODE ANCHOR %s %s: |device f16 - exact| max %.2e p99 %.2e median %.2e px | |oracle fp32 - exact| max %.2e p99 %.2e "
              "median %.2e px | max |flow| %.1f px | ratio of medians %.0f" % ((world, name) + d + o + (mag, d[2] / max(o[2], 1e-12))))
        # one f16 rounding per operand is 2^-11 relative; through ~60 layers and the data-dependent warps of the coarse-to-fine
        # scheme the flow moves by a small fraction of a pixel.  Gate (3x the first measured run, see profiles/r5_f16_mode.txt):
        assert d[2] <= F16_ANCHOR_GATE[world][0] and d[1] <= F16_ANCHOR_GATE[world][1], (world, name, d)


# (median, p99) px of |flow - anchor|, see the test above
F16_ANCHOR_GATE = {"random_weights_192x640": (0.05, 0.5), "coded_tunnel_256x640": (0.05, 0.5)}


def test_f16_trajectory_report(gpu, f16_mode, tmp_path):
    pmod = importlib.import_module("df-vo_amd.pipeline")
    smod = importlib.import_module("df-vo_amd.sequence")
    fx = np.load(GOLD)
    h, w, n = int(fx["h"]), int(fx["w"]), int(fx["n_frames"])
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=21)
    pipe = pmod.TrackingPipeline(h, w, 192, 640, seq["K"], crafted_liteflownet_state_dict(h, w, "mux"),
                                 crafted_monodepth2_state_dict(), seed=4869)
    frames = smod.frames_to_device(seq["frames"])
    kps, modes = [], []

    def collect(j, out):
        modes.append(int(out.status))
        kps.append(pipe.get_keypoints(j % smod.SLOTS))

    poses, gathered = smod.run_sequence(pipe, frames, n, collect=collect)
    pipe.close()
    off = np.concatenate([[0], np.cumsum(fx["n_kp"])])
    overlap, dF, nkp = [], [], []
    rel = gathered[:, :16].reshape(-1, 4, 4)
    for j in range(n - 1):
        kr = kps[j][0]
        xy = fx["kp_xy"][off[j]:off[j + 1]].astype(np.int64)
        a = set(map(tuple, kr.astype(np.int64)))
        overlap.append(len(a & set(map(tuple, xy))) / max(1, len(a)))
        nkp.append(len(kr))
        dF.append(np.linalg.norm(rel[j] - fx["seq_rel"][j]))
    dF = np.array(dF)
    gt = list(seq["poses"])
    ev = E.evaluate(gt, list(poses))
    ev_o = {k[5:]: float(fx[k]) for k in fx.files if k.startswith("eval_")}
    same_branch = sum((modes[j] == 0) == (str(fx["seq_status"][j]) == "E") for j in range(n - 1))
    print("F16-MODE TRAJECTORY 130-frame tunnel, sequential RandomState: t_rel %.4f %% (fp32 oracle %.4f, delta %+.4f)  r_rel %.4f "
          "deg/100m (oracle %.4f)  ATE %.3f m (oracle %.3f)  RPE %.4f m / %.4f deg (oracle %.4f / %.4f) | E %d PnP %d const %d, "
          "same tracking branch as the oracle on %d of %d pairs | keypoints per pair median %d, shared with the oracle's as a "
          "set: median %.1f %% min %.1f %% | ||dT||_F vs the oracle's pose: median %.2e max %.2e" % (
              ev["t_rel"], ev_o["t_rel"], ev["t_rel"] - ev_o["t_rel"], ev["r_rel"], ev_o["r_rel"], ev["ate"], ev_o["ate"],
              ev["rpe_t"], ev["rpe_r"], ev_o["rpe_t"], ev_o["rpe_r"], modes.count(0), modes.count(3), modes.count(1),
              same_branch, n - 1, int(np.median(nkp)), 100 * np.median(overlap), 100 * min(overlap), np.median(dF), dF.max()))
    assert (gathered[:, 16] != 2).all()
    assert ev["t_rel"] < 2.0, "the f16-mode tracker must still follow the rendered camera"
    assert abs(ev["t_rel"] - ev_o["t_rel"]) <= 0.5  # (the fp32-class modes are gated at 0.1; measured delta: see the printed line)
    assert np.median(overlap) >= 0.5
