"""Fixtures computed by the ORACLE (not by the reference) in the build container, for cases the GPU box cannot afford to
recompute inside the -m gpu suite:

  coded_large.npz         see coded_large() below
  tunnel_traj.npz         see tunnel_trajectory() below
  flownet_1280x1920.npz   BASELINE config 5: torch-CPU LiteFlowNet (oracle/nets_torch.py, pinned to the reference's own
                          LiteFlowNet class by liteflownet_64x96.npz) on the seeded 1280x1920 pair of
                          tests/test_nets_gpu.py::test_flownet_large_configs; forward / backward flow and the consistency
                          map sub-sampled every 8th pixel + CRC32 of the full maps and of the two input frames

    python tests/golden/make_oracle_fixtures.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import nets_torch as O  # noqa: E402
from synth import image_pair  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def flownet_large(h=1280, w=1920, step=8):
    sd = O.liteflownet_state_dict(4869)
    ref_img, cur_img = image_pair(h, w, seed=2000 + h)
    fwd, bwd, diff = O.flow_inference(sd, ref_img, cur_img)
    np.savez_compressed(os.path.join(HERE, "flownet_%dx%d.npz" % (h, w)), step=step, fwd=fwd[:, ::step, ::step],
                        bwd=bwd[:, ::step, ::step], diff=diff[::step, ::step, 0], img_crc=np.array([crc(ref_img), crc(cur_img)]),
                        fwd_absmax=np.abs(fwd).max(), bwd_absmax=np.abs(bwd).max())
    print("flownet %dx%d: |fwd| max %.2f, fixture written" % (h, w, np.abs(fwd).max()))


def coded_large(sizes=((960, 1280), (1280, 1920)), n_frames=3, step=8):
    """coded_large.npz: BASELINE configs 4 / 5 frame sizes through the image-level path -- the torch-CPU LiteFlowNet oracle
    on consecutive coded tunnel frames (tests/test_e2e_gpu.py: 'mux' encoding, forward drive), forward / backward flow and
    consistency map every 8th pixel, frames pinned by CRC.  The -m gpu test compares the HIP nets with these and runs the
    solver-stage oracle live on the device's own arrays."""
    import importlib
    import time
    syn = importlib.import_module("df-vo_amd.synthetic")
    out = {"step": step, "n_frames": n_frames}
    for h, w in sizes:
        seq = syn.coded_tunnel_sequence(h, w, n_frames, mode="mux", step=1.0)
        fsd = syn.crafted_liteflownet_state_dict(h, w, "mux")
        out["crc_%dx%d" % (h, w)] = crc(seq["frames"])
        for k in range(n_frames - 1):
            t0 = time.time()
            O._grid_cache.clear()
            fwd, bwd, diff = O.flow_inference(fsd, seq["frames"][k], seq["frames"][k + 1])
            key = "%dx%d_%d" % (h, w, k)
            out["fwd_" + key], out["bwd_" + key] = fwd[:, ::step, ::step], bwd[:, ::step, ::step]
            out["diff_" + key] = diff[::step, ::step, 0]
            print("coded %dx%d pair %d: |fwd| max %.2f (%.0f s)" % (h, w, k, np.abs(fwd).max(), time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "coded_large.npz"), **out)


def tunnel_trajectory(h=256, w=640, n_frames=130):
    """tunnel_traj.npz: the oracle's frame loop (oracle/pipeline_np.py) over the coded tunnel sequence of
    tests/test_trajectory_gpu.py -- global poses, tracking modes, the trajectory metrics of oracle/kitti_eval.py against
    the rendered ground truth, CRC of the frames.  Round 3: also the per-pair record the from-images accounting needs
    (north_star: "bit-exact RANSAC inlier masks under a fixed seed, pose within 1e-4 Frobenius on identical image pairs"):
    keypoints (integer pixel of the reference keypoint + its float32 flow: kp_cur = kp_ref + flow exactly as
    keypoint_sampler.py:103 forms it), the E-tracker's inlier mask and the relative pose of every pair, once under the
    reference's sequential RandomState (apis/run.py:81-84) and once re-seeded per pair with seed ^ (pair + 1) (the
    data-parallel mode of df-vo_amd/sequence.py, in which a pair's result does not depend on the pairs before it)."""
    import importlib
    from oracle import kitti_eval as E
    from oracle import pipeline_np as P
    syn = importlib.import_module("df-vo_amd.synthetic")
    seq = syn.coded_tunnel_sequence(h, w, n_frames, mode="mux", step=1.0, seed=21)
    fsd, dsd = syn.crafted_liteflownet_state_dict(h, w, "mux"), syn.crafted_monodepth2_state_dict()
    frames, K, seed = list(seq["frames"]), seq["K"], 4869
    np.random.seed(seed)
    st_seq = np.random.get_state()
    rec = {m: {"rel": [], "status": [], "kp_xy": [], "kp_flow": [], "mask": [], "n_kp": []} for m in ("seq", "pp")}
    _, depth_ref = P.frame_depth(dsd, frames[0])
    prev = {"seq": np.eye(4), "pp": np.eye(4)}
    g = np.eye(4)
    poses = [g.copy()]
    for k in range(1, n_frames):
        _, depth_cur = P.frame_depth(dsd, frames[k])
        fwd, bwd, diff = O.flow_inference(fsd, frames[k - 1], frames[k])
        for m in ("seq", "pp"):
            if m == "seq":
                np.random.set_state(st_seq)
            else:
                np.random.seed((seed ^ k) & 0xffffffff)  # pair index j = k - 1: seed ^ (j + 1)
            r = P.solve_pair(fwd, diff, depth_cur, depth_ref, K)
            if m == "seq":
                st_seq = np.random.get_state()
            rel = prev[m].copy() if r["pose"] is None else r["pose"]
            prev[m] = rel
            d = rec[m]
            d["rel"].append(rel)
            d["status"].append(r["status"])
            if r.get("kp_ref") is not None:
                xy = r["kp_ref"].astype(np.int64)
                assert np.array_equal(xy.astype(np.float64), r["kp_ref"])
                fl = fwd[:, xy[:, 1], xy[:, 0]].T.astype(np.float32)
                assert np.array_equal(r["kp_ref"] + fl.astype(np.float64), r["kp_cur"])
                d["kp_xy"].append(xy.astype(np.int16))
                d["kp_flow"].append(fl)
                d["mask"].append(np.asarray(r["E"]["inliers"]).reshape(-1).astype(np.uint8))
                d["n_kp"].append(len(xy))
            else:
                d["n_kp"].append(0)
            if m == "seq":
                g = P.update_global_pose(g, rel)
                poses.append(g.copy())
        depth_ref = depth_cur
        if k % 10 == 0:
            print("  frame %d seq %s pp %s" % (k, rec["seq"]["status"][-1], rec["pp"]["status"][-1]), flush=True)
    poses = np.stack(poses)
    ev = E.evaluate(list(seq["poses"]), list(poses))
    st = rec["seq"]["status"]
    print("oracle trajectory:", ev, "modes", {m: st.count(m) for m in set(st)})
    cat = lambda xs, dt, shp: (np.concatenate(xs, 0) if xs else np.zeros(shp, dt))
    d = rec["seq"]  # keypoint selection draws nothing from the RandomState: one copy serves both modes
    assert all(np.array_equal(a, b) for a, b in zip(d["kp_xy"], rec["pp"]["kp_xy"])) and d["n_kp"] == rec["pp"]["n_kp"]
    extra = {"n_kp": np.array(d["n_kp"]), "kp_xy": cat(d["kp_xy"], np.int16, (0, 2)), "kp_flow": cat(d["kp_flow"], np.float32, (0, 2))}
    for m, d in rec.items():
        extra.update({m + "_rel": np.stack(d["rel"]), m + "_status": np.array(d["status"]),
                      m + "_mask": np.packbits(cat(d["mask"], np.uint8, (0,)))})
    np.savez_compressed(os.path.join(HERE, "tunnel_traj.npz"), poses=poses, gt=seq["poses"],
                        status=np.array(st), frames_crc=crc(seq["frames"]), n_frames=n_frames, h=h, w=w,
                        **{"eval_" + k: v for k, v in ev.items()}, **extra)


if __name__ == "__main__":
    what = sys.argv[1:] or ["flownet_large", "tunnel_trajectory"]
    if "flownet_large" in what:
        flownet_large()
    if "tunnel_trajectory" in what:
        tunnel_trajectory()
    if "coded_large" in what:
        coded_large()
