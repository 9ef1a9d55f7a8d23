"""ORACLE (test infrastructure only -- never imported by the product path).

Image-level restatement of one DF-VO tracking step and of the frame loop around it, on the torch-CPU nets
(oracle/nets_torch.py) and the numpy / C solver oracle (oracle/tracker_np.py).  Follows (paths relative to
/root/reference):
    libs/dfvo.py:299-345      deep_model_inference: forward_depth on the current frame (PIL LANCZOS to the feed size,
                              deep_models.py:184-206), cv2.resize INTER_NEAREST to the image size, preprocess_depth;
                              forward_flow(cur, ref, forward_backward=True)
    libs/dfvo.py:121-262      tracking, tracking_method 'hybrid': kp selection, E-tracker, scale recovery, PnP fallback,
                              constant-motion model when no good keypoints are found
    libs/dfvo.py:109-119      update_global_pose
    libs/dfvo.py:347-425      main loop (frame 0 only gets its depth)
Pinning: every stage called here is pinned by its own fixture (tests/golden/*.npz, see tracker_np.py / nets_torch.py);
the loop itself is checked against the reference's unmodified DFVO.tracking / update_global_pose in
tests/golden/make_golden.py (fixture dfvo_tracking.npz).
"""
import numpy as np
from PIL import Image

from . import cv2_shim
from . import nets_torch as O
from . import tracker_np as T

DEPTH_CROP = [[0.3, 1], [0, 1]]  # options/examples/default_configuration.yml crop.depth_crop


def frame_depth(dsd, img_u8, feed_hw=(192, 640), depth_range=(0.0, 50.0), dtype=None):
    """dfvo.py:305-319: raw_depth (float32, image size) and depth (float64, cropped / range-masked) of one frame"""
    h, w = img_u8.shape[:2]
    feed = np.asarray(Image.fromarray(img_u8).resize((feed_hw[1], feed_hw[0]), Image.LANCZOS))  # deep_models.py:195-199
    small = O.depth_inference(dsd, np.ascontiguousarray(feed), **({"dtype": dtype} if dtype is not None else {}))
    small = small.astype(np.float32)  # (float64 anchor runs: rounded once, where the reference's net output is float32)
    raw = cv2_shim.resize(small, (w, h), interpolation=cv2_shim.INTER_NEAREST)
    return raw, T.preprocess_depth(raw, DEPTH_CROP, list(depth_range))


def solve_pair(fwd, diff, depth_cur, depth_ref, K, num_bestN=2000, e_max_iters=1000):
    """dfvo.py:147-250 on given net outputs.  Consumes np.random exactly as the reference does.
    `num_bestN` = kp_selection.local_bestN.num_bestN of the configuration (kp_selection.py:74-200); `e_max_iters` = the
    findEssentialMat hypothesis budget (1000 is OpenCV 3.4.3's fixed value; BASELINE config 5 asks for 8192).
    Returns dict(status 'E' | 'PnP' | 'constant_motion', pose 4x4 cur->ref or None, + stage results for bit comparisons)."""
    if diff.ndim == 2:
        diff = diff[..., None]
    out = {"status": "constant_motion", "pose": None}
    kp = T.local_bestN(fwd, diff, num_bestN=num_bestN)
    out["good_kp_found"] = bool(kp["good_kp_found"])
    if not kp["good_kp_found"]:
        return out
    kp1, kp2 = kp["kp1_best"][0], kp["kp2_best"][0]
    out["kp_ref"], out["kp_cur"] = kp1, kp2
    res = T.compute_pose_2d2d(kp1, kp2, K, max_iters=e_max_iters)
    out["E"] = res
    scale = -1
    if np.linalg.norm(res["t"]) != 0:  # dfvo.py:184-222
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3:] = res["R"], res["t"]
        diag = {}
        scale = T.find_scale_from_depth(kp1, kp2, np.linalg.inv(pose), depth_cur, K, diag=diag)
        out["scale"], out["scale_diag"] = scale, diag
    if np.linalg.norm(res["t"]) == 0 or scale == -1:  # dfvo.py:225-250
        pnp = T.compute_pose_3d2d(kp1, kp2, depth_ref, K)
        out["pnp"] = pnp
        out["status"], out["pose"] = "PnP", pnp["pose"]
        return out
    pose = np.eye(4)
    pose[:3, :3] = res["R"]
    pose[:3, 3:] = res["t"] * scale
    out["status"], out["pose"] = "E", pose
    return out


def update_global_pose(g, rel):
    """dfvo.py:109-119 with scale 1"""
    n = g.copy()
    n[:3, 3:] = g[:3, :3] @ rel[:3, 3:] + g[:3, 3:]
    n[:3, :3] = g[:3, :3] @ rel[:3, :3]
    return n


def track_sequence(frames_u8, fsd, dsd, K, seed=4869, feed_hw=(192, 640), progress=None):
    """the reference's frame loop on the oracle: returns dict(poses [n,4,4] global camera-to-world, rel [n-1,4,4],
    status list)"""
    np.random.seed(seed)  # apis/run.py:81-84
    n = len(frames_u8)
    g = np.eye(4)
    poses = [g.copy()]
    rels, status = [], []
    _, depth_ref = frame_depth(dsd, frames_u8[0], feed_hw)
    prev = np.eye(4)
    for k in range(1, n):
        _, depth_cur = frame_depth(dsd, frames_u8[k], feed_hw)
        fwd, bwd, diff = O.flow_inference(fsd, frames_u8[k - 1], frames_u8[k])
        r = solve_pair(fwd, diff, depth_cur, depth_ref, K)
        rel = prev.copy() if r["pose"] is None else r["pose"]  # constant motion: dfvo.py:157-161
        g = update_global_pose(g, rel)
        poses.append(g.copy())
        rels.append(rel)
        status.append(r["status"])
        prev = rel
        depth_ref = depth_cur
        if progress:
            progress(k, r)
    return {"poses": np.stack(poses), "rel": np.stack(rels) if rels else np.zeros((0, 4, 4)), "status": status}
