"""Oracle-vs-oracle from-images spread.

DESIGN.md section 4 attributes the device's from-images gap (0 of 129 pairs with an identical keypoint list against the
torch-CPU oracle run from the same uint8 frames) to the keypoint selection ranking a consistency map whose smallest values
are rounding noise: ANY change of the fp32 summation order inside the nets reshuffles the ranking.  This script measures
that claim on the oracle alone: the 130-frame coded tunnel sequence of tests/golden/tunnel_traj.npz is run again through
the SAME oracle code (oracle/pipeline_np.py: torch-CPU nets + C/numpy solvers) with a different CPU execution of the same
fp32 convolutions -- one thread instead of all cores, and / or oneDNN off (torch's native convolution) -- and compared
with the committed fixture (all cores, oneDNN) exactly as tests/test_trajectory_gpu.py compares the device.

    python tools/oracle_thread_spread.py [--threads 1] [--no-mkldnn] [--frames 130] > profiles/r4_oracle_vs_oracle.txt

--float64 runs both nets in DOUBLE on the same fp32 inputs / weights / grid constants (oracle/nets_torch.py, dtype=) and
rounds their outputs to float32 once: the exact function the fp32 executions approximate.  Compared with the fp32 fixture
it says how far the REFERENCE's own arithmetic is from its function -- the yardstick for the device's distance.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets_torch as O  # noqa: E402
from oracle import pipeline_np as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--no-mkldnn", action="store_true")
    ap.add_argument("--frames", type=int, default=130)
    ap.add_argument("--float64", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    if a.no_mkldnn:
        torch.backends.mkldnn.enabled = False
    fx = np.load(os.path.join(ROOT, "tests", "golden", "tunnel_traj.npz"))
    h, w, n = int(fx["h"]), int(fx["w"]), min(int(fx["n_frames"]), a.frames)
    syn = importlib.import_module("df-vo_amd.synthetic")
    seq = syn.coded_tunnel_sequence(h, w, int(fx["n_frames"]), mode="mux", step=1.0, seed=21)
    fsd, dsd = syn.crafted_liteflownet_state_dict(h, w, "mux"), syn.crafted_monodepth2_state_dict()
    frames, K, seed = list(seq["frames"]), seq["K"], 4869
    off = np.concatenate([[0], np.cumsum(fx["n_kp"])])
    np.random.seed(seed)
    st_seq = np.random.get_state()
    res = {m: dict(same_kp=0, same_mask=0, same_cnt=0, dF=[], overlap=[], moved=[], flow_diff=[]) for m in ("seq", "pp")}
    mask_bits = {m: np.unpackbits(fx[m + "_mask"]) for m in ("seq", "pp")}
    dt = torch.float64 if a.float64 else torch.float32
    _, depth_ref = P.frame_depth(dsd, frames[0], dtype=dt)
    t0 = time.time()
    for k in range(1, n):
        _, depth_cur = P.frame_depth(dsd, frames[k], dtype=dt)
        fwd, bwd, diff = (x.astype(np.float32) for x in O.flow_inference(fsd, frames[k - 1], frames[k], dtype=dt))
        j = k - 1
        xy0 = fx["kp_xy"][off[j]:off[j + 1]].astype(np.float64)
        cur0 = xy0 + fx["kp_flow"][off[j]:off[j + 1]].astype(np.float64)
        for m in ("seq", "pp"):
            if m == "seq":
                np.random.set_state(st_seq)
            else:
                np.random.seed((seed ^ k) & 0xffffffff)
            r = P.solve_pair(fwd, diff, depth_cur, depth_ref, K)
            if m == "seq":
                st_seq = np.random.get_state()
            d = res[m]
            rel = r["pose"] if r["pose"] is not None else np.eye(4)
            d["dF"].append(np.linalg.norm(rel - fx[m + "_rel"][j]))
            kr, kc = r.get("kp_ref"), r.get("kp_cur")
            if kr is None:
                continue
            d["same_cnt"] += len(kr) == len(xy0)
            if len(kr) == len(xy0):
                d["moved"].append(int((kr != xy0).any(1).sum()))
            a_ = set(map(tuple, kr.astype(np.int64)))
            d["overlap"].append(len(a_ & set(map(tuple, xy0.astype(np.int64)))) / max(1, len(a_)))
            same = len(kr) == len(xy0) and np.array_equal(kr, xy0) and np.array_equal(kc, cur0)
            d["same_kp"] += same
            inl = np.asarray(r["E"]["inliers"]).reshape(-1).astype(bool)
            d["same_mask"] += bool(same and np.array_equal(inl, mask_bits[m][off[j]:off[j + 1]].astype(bool)))
            if m == "seq":  # how far the two CPU executions of the same fp32 net are apart, at the fixture's keypoints
                xi = xy0.astype(np.int64)
                fl = fwd[:, xi[:, 1], xi[:, 0]].T
                d["flow_diff"].append(float(np.abs(fl - fx["kp_flow"][off[j]:off[j + 1]]).max()))
        depth_ref = depth_cur
        if k % 10 == 0:
            sys.stderr.write("  frame %d (%.0f s)\n" % (k, time.time() - t0))
    print("ORACLE vs ORACLE, from the same uint8 frames: this run = torch %s, %d thread(s), oneDNN %s, nets in %s; fixture = all cores, oneDNN on, float32"
          % (torch.__version__, a.threads, "off" if a.no_mkldnn else "on", "FLOAT64 (outputs rounded to float32)" if a.float64 else "float32"))
    for m, name in (("seq", "sequential"), ("pp", "per_pair")):
        d = res[m]
        dF = np.array(d["dF"])
        print("FROM-IMAGES %d-frame %s RandomState (oracle, other CPU execution): of %d pairs | identical keypoint set (values+order) %d | "
              "identical inlier mask %d | same keypoint count %d, list positions holding another pixel: median %d max %d of ~2000, "
              "keypoints shared as a set: median %.1f %% min %.1f %% | ||dT||_F <= 1e-4: %d, <= 1e-3: %d, <= 1e-2: %d; median %.2e max %.2e"
              % (n, name, n - 1, d["same_kp"], d["same_mask"], d["same_cnt"], np.median(d["moved"]) if d["moved"] else -1,
                 max(d["moved"]) if d["moved"] else -1, 100 * np.median(d["overlap"]), 100 * min(d["overlap"]),
                 (dF <= 1e-4).sum(), (dF <= 1e-3).sum(), (dF <= 1e-2).sum(), np.median(dF), dF.max()))
    fd = res["seq"]["flow_diff"]
    print("max |flow(this run) - flow(fixture)| at the fixture's keypoints: median over pairs %.2e px, max %.2e px" % (np.median(fd), max(fd)))


if __name__ == "__main__":
    main()
