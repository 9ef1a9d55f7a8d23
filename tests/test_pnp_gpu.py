"""GPU parity of the PnP fallback (dfvo_compute_pose_3d2d and the PnpTracker mirror) against the oracle
(oracle/tracker_np.compute_pose_3d2d over oracle/cv3_pnp.c) on identical inputs and an identical numpy RandomState.
Bit-exact: the surviving keypoints, the best inlier count, rvec-derived R and t, the RandomState after the call.
The unprojected points are compared after the float32 conversion solvePnPRansac applies (numpy's batched matmul may
contract to FMA; the device sums left to right)."""
import ctypes as C
import importlib

import numpy as np
import pytest

from oracle import tracker_np as T

pytestmark = pytest.mark.gpu

K = np.array([[718.856, 0, 607.19], [0, 718.856, 185.22], [0, 0, 1.0]])


@pytest.fixture(scope="module")
def trk(gpu):
    lib = gpu.lib()
    t = C.c_void_p()
    gpu.check(lib.dfvo_tracker_create(None, C.byref(t)))
    yield t
    lib.dfvo_tracker_destroy(t)


def np_state():
    st = np.random.get_state()
    return np.ascontiguousarray(np.r_[st[1].astype(np.uint32), np.uint32(st[2])])


def pnp_case(seed, h=376, w=1241, n=1500, outliers=0.25, hole_frac=0.1):
    """keypoints on the pixel grid of view 1 with a depth map, their noisy reprojection into view 2"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    depth = np.clip(4.0 + 60.0 * (1.0 - yy / h) + 3.0 * np.sin(xx / 37.0) + 2.0 * np.cos(yy / 23.0), 0.5, 80.0)
    depth[rng.random((h, w)) < hole_frac] = 0.0  # invalid depth
    depth[:int(0.1 * h)] = 70.0                  # beyond max_depth
    pick = rng.choice(h * w, n, replace=False)
    kp1 = np.stack([pick % w, pick // w], 1).astype(np.float64)
    d = depth[kp1[:, 1].astype(int), kp1[:, 0].astype(int)]
    d_safe = np.where(d > 0, d, 10.0)
    X = np.stack([(kp1[:, 0] - K[0, 2]) / K[0, 0] * d_safe, (kp1[:, 1] - K[1, 2]) / K[1, 1] * d_safe, d_safe], 1)
    rv = np.array([0.003, -0.012, 0.002]) * rng.uniform(0.5, 2)
    th = np.linalg.norm(rv)
    kx = rv / th
    Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.03, -0.01, 0.9]) * rng.uniform(0.5, 1.5)
    Xc = X @ R.T + t
    kp2 = np.stack([Xc[:, 0] / Xc[:, 2] * K[0, 0] + K[0, 2], Xc[:, 1] / Xc[:, 2] * K[1, 1] + K[1, 2]], 1)
    kp2 += rng.normal(0, 0.2, kp2.shape)
    out = rng.random(n) < outliers
    kp2[out] += rng.normal(0, 30, (int(out.sum()), 2))  # some of these leave the image
    return np.ascontiguousarray(kp1), np.ascontiguousarray(kp2), np.ascontiguousarray(depth), R, t


def run_hip(gpu, trk, kp1, kp2, depth, repeat=5, iters=100, thre=1.0, min_depth=0.0, max_depth=50.0):
    capi = gpu
    cfg = capi.Pose3d2dCfg(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], min_depth=min_depth, max_depth=max_depth,
                           repeat=repeat, iters=iters, reproj_thre=thre)
    Kinv = np.linalg.inv(K)
    for i in range(9):
        cfg.Kinv[i] = Kinv.flat[i]
    out = capi.Pose3d2dOut()
    keep = np.zeros(max(len(kp1), 1), np.uint8)
    s = np_state()
    capi.check(capi.lib().dfvo_tracker_set_rng_state(trk, capi.as_ptr(s)))
    h, w = depth.shape
    capi.check(capi.lib().dfvo_compute_pose_3d2d(trk, capi.as_ptr(kp1), capi.as_ptr(kp2), len(kp1), capi.as_ptr(depth), h, w,
                                                 C.byref(cfg), C.byref(out), capi.as_ptr(keep)))
    s2 = np.zeros(625, np.uint32)
    capi.check(capi.lib().dfvo_tracker_get_rng_state(trk, capi.as_ptr(s2)))
    return out, keep[:len(kp1)].astype(bool), s2


@pytest.mark.parametrize("seed,n,repeat", [(21, 1500, 5), (22, 2000, 5), (23, 300, 3), (24, 40, 5)])
def test_compute_pose_3d2d_bit_exact(gpu, trk, seed, n, repeat):
    kp1, kp2, depth, R_true, t_true = pnp_case(seed, n=n)
    np.random.seed(4869 + seed)
    out, keep, state_after = run_hip(gpu, trk, kp1, kp2, depth, repeat=repeat)
    np.random.seed(4869 + seed)
    ref = T.compute_pose_3d2d(kp1, kp2, depth, K, 0.0, 50.0, repeat, 100, 1.0)
    assert out.n_filtered == len(ref["kp1"])
    assert np.array_equal(kp1[keep], ref["kp1"]) and np.array_equal(kp2[keep], ref["kp2"])
    assert out.found == 1 and out.best_inliers == ref["best_inlier"]
    assert np.array_equal(np.array(out.R[:]).reshape(3, 3), ref["R"])          # bit-exact
    assert np.array_equal(np.array(out.tvec[:]).reshape(3, 1), ref["t"])
    assert np.array_equal(state_after, np_state())                              # the shuffles left the same stream
    # and the pose is the right one
    assert np.abs(ref["R"] - R_true).max() < 2e-3 and np.abs(ref["t"].ravel() - t_true).max() < 5e-2


def test_too_few_points_gives_identity_but_consumes_the_stream(gpu, trk):
    kp1, kp2, depth, _, _ = pnp_case(31, n=200)
    depth2 = depth.copy()
    depth2[:] = 0.0
    pick = kp1[:4].astype(int)
    depth2[pick[:, 1], pick[:, 0]] = 10.0  # only four keypoints keep a valid depth
    kp2c = kp2.copy()
    kp2c[:4] = np.clip(kp2c[:4], 1, 300)
    np.random.seed(77)
    out, keep, state_after = run_hip(gpu, trk, kp1, kp2c, depth2)
    np.random.seed(77)
    ref = T.compute_pose_3d2d(kp1, kp2c, depth2, K, 0.0, 50.0, 5, 100, 1.0)
    assert out.found == 0 and out.n_filtered == len(ref["kp1"]) == 4 and ref["best_inlier"] == 0
    assert np.array_equal(np.array(out.R[:]).reshape(3, 3), np.eye(3))
    assert np.array_equal(state_after, np_state())
    # no keypoints at all
    np.random.seed(78)
    out, keep, state_after = run_hip(gpu, trk, np.zeros((0, 2)), np.zeros((0, 2)), depth)
    assert out.found == 0 and out.n_filtered == 0
    np.random.seed(78)
    assert np.array_equal(state_after, np_state())


def _cfg():
    class NS(dict):
        __getattr__ = dict.__getitem__
    return NS(kp_selection=NS(rigid_flow_kp=NS(enable=False)), depth=NS(min_depth=0.0, max_depth=50.0),
              pnp_tracker=NS(ransac=NS(iter=100, reproj_thre=1.0, repeat=5)))


def test_pnp_tracker_class_pose(gpu):
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    trk_mod = importlib.import_module("df-vo_amd.libs.tracker")
    cam = cam_mod.Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
    tracker = trk_mod.PnpTracker(_cfg(), cam)
    kp1, kp2, depth, R_true, t_true = pnp_case(41, n=1200)
    np.random.seed(5)
    res = tracker.compute_pose_3d2d(kp1, kp2, depth, True)
    state_hip = np_state()
    np.random.seed(5)
    ref = T.compute_pose_3d2d(kp1, kp2, depth, K, 0.0, 50.0, 5, 100, 1.0)
    assert np.array_equal(state_hip, np_state())  # the global numpy stream advanced exactly as in the reference
    assert np.abs(res["pose"].pose - ref["pose"]).max() <= 1e-12  # Frobenius bar of the contract is 1e-4
    assert np.array_equal(res["kp1"], ref["kp1"]) and np.array_equal(res["kp2"], ref["kp2"])


def planar_case(seed, h=376, w=1241, n=1200, relief=0.02):
    """keypoints whose valid depths lie on the ground plane up to `relief` (relative; a road-only view): the inlier object
    points of solvePnPRansac are coplanar in cvFindExtrinsicCameraParams2's sense (W[2] / W[1] ~ 1e-6 << 1e-3) and it takes
    its planar branch.  With relief = 0 the five-point EPnP hypotheses themselves degenerate (OpenCV does not special-case
    coplanar control points): the consensus set is a handful of points and the pose meaningless."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    with np.errstate(divide="ignore"):
        depth = np.where(yy > K[1, 2] + 12, 1.65 * K[1, 1] / (yy - K[1, 2]), 0.0)  # plane Y = 1.65
    depth = depth * (1.0 + relief * np.sin(xx / 40.0) * np.cos(yy / 9.0))
    pick = rng.choice(h * w, n, replace=False)
    kp1 = np.stack([pick % w, pick // w], 1).astype(np.float64)
    d = depth[kp1[:, 1].astype(int), kp1[:, 0].astype(int)]
    d_safe = np.where(d > 0, d, 10.0)
    X = np.stack([(kp1[:, 0] - K[0, 2]) / K[0, 0] * d_safe, (kp1[:, 1] - K[1, 2]) / K[1, 1] * d_safe, d_safe], 1)
    rv = np.array([0.002, -0.01, 0.001])
    th = np.linalg.norm(rv)
    kx = rv / th
    Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.02, -0.01, 0.7])
    Xc = X @ R.T + t
    kp2 = np.stack([Xc[:, 0] / Xc[:, 2] * K[0, 0] + K[0, 2], Xc[:, 1] / Xc[:, 2] * K[1, 1] + K[1, 2]], 1)
    kp2 += rng.normal(0, 0.15, kp2.shape)
    return np.ascontiguousarray(kp1), np.ascontiguousarray(kp2), np.ascontiguousarray(depth), R, t


@pytest.mark.parametrize("seed", [51, 52])
def test_compute_pose_3d2d_coplanar_object_points(gpu, trk, seed):
    """the planar case: cvFindExtrinsicCameraParams2's homography initialisation (round 3: restated on both sides --
    oracle/cv3_pnp.c cv3_find_extrinsic_guess, k_pnp_refine) -- device == oracle BIT FOR BIT (the refinement no longer
    consumes the RANSAC model's last bits, only its inlier mask), and the pose is the right one"""
    kp1, kp2, depth, R_true, t_true = planar_case(seed)
    np.random.seed(99 + seed)
    out, keep, state_after = run_hip(gpu, trk, kp1, kp2, depth)
    np.random.seed(99 + seed)
    ref = T.compute_pose_3d2d(kp1, kp2, depth, K, 0.0, 50.0, 5, 100, 1.0)
    assert out.found == 1 and out.status != -2 and out.best_inliers == ref["best_inlier"] > 400
    assert np.array_equal(np.array(out.R[:]).reshape(3, 3), ref["R"])
    assert np.array_equal(np.array(out.tvec[:]).reshape(3, 1), ref["t"])
    assert np.array_equal(state_after, np_state())
    assert np.abs(ref["R"] - R_true).max() < 2e-3 and np.abs(ref["t"].ravel() - t_true).max() < 5e-2
    # exactly coplanar points: degenerate EPnP hypotheses, a meaningless pose -- but no abort, and the same consensus size
    kp1, kp2, depth, _, _ = planar_case(seed, relief=0.0)
    np.random.seed(7 + seed)
    out, keep, state_after = run_hip(gpu, trk, kp1, kp2, depth)
    np.random.seed(7 + seed)
    ref = T.compute_pose_3d2d(kp1, kp2, depth, K, 0.0, 50.0, 5, 100, 1.0)
    assert out.status != -2 and out.n_filtered == len(ref["kp1"]) and np.array_equal(state_after, np_state())


def test_compute_pose_3d2d_depth_at_keypoints_entry(gpu, trk):
    """dfvo_compute_pose_3d2d_at_kp (the depth map's values at kp1's pixels instead of the H x W map, RandomState in and out of the
    same call -- what the PnpTracker mirror uses) returns what dfvo_compute_pose_3d2d returns, bit for bit; keypoints with
    negative (wrapping) / out-of-map / non-finite kp1 coordinates and out-of-image kp2 included"""
    capi = gpu
    for seed, n in ((31, 1500), (32, 60)):
        kp1, kp2, depth, _, _ = pnp_case(seed, n=n)
        kp1, kp2 = kp1.copy(), kp2.copy()
        h, w = depth.shape
        kp1[2] = [-3.5, 20.0]            # wraps to column w - 3
        kp1[4] = [30.0, -2.25]           # wraps to row h - 2
        kp1[6] = [-(w + 5.0), 8.0]       # still negative after one wrap: dropped
        kp1[8] = [w + 1.5, 8.0]          # outside: dropped
        kp1[10] = [np.nan, 8.0]
        kp2[12] = [-0.5, 10.0]           # kp2 outside the image: dropped before depth is looked at
        np.random.seed(4869 + seed)
        o0, keep0, st0 = run_hip(gpu, trk, kp1, kp2, depth)
        xi, yi = np.trunc(kp1[:, 0]), np.trunc(kp1[:, 1])
        with np.errstate(invalid="ignore"):
            xi, yi = np.where(xi < 0, xi + w, xi), np.where(yi < 0, yi + h, yi)
            inside = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
        at_kp = np.full(len(kp1), 3.0)   # (a valid-looking depth at dropped keypoints must not matter)
        at_kp[inside] = depth[yi[inside].astype(int), xi[inside].astype(int)]
        cfg = capi.Pose3d2dCfg(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], min_depth=0.0, max_depth=50.0, repeat=5, iters=100,
                               reproj_thre=1.0)
        Kinv = np.linalg.inv(K)
        for i in range(9):
            cfg.Kinv[i] = Kinv.flat[i]
        o1 = capi.Pose3d2dOut()
        keep1 = np.zeros(len(kp1), np.uint8)
        np.random.seed(4869 + seed)
        rng = np_state()
        capi.check(capi.lib().dfvo_compute_pose_3d2d_at_kp(trk, capi.as_ptr(kp1), capi.as_ptr(kp2), len(kp1), capi.as_ptr(at_kp), h, w,
                                                           C.byref(cfg), capi.as_ptr(rng), C.byref(o1), capi.as_ptr(keep1)))
        print("n=%d map entry: found %d inliers %d filtered %d | per-keypoint entry: found %d inliers %d filtered %d" % (
            n, o0.found, o0.best_inliers, o0.n_filtered, o1.found, o1.best_inliers, o1.n_filtered))
        assert (o0.found, o0.best_inliers, o0.n_filtered, o0.status) == (o1.found, o1.best_inliers, o1.n_filtered, o1.status)
        assert o0.n_filtered < len(kp1) and not keep0[[6, 8, 10, 12]].any()
        assert np.array_equal(keep0, keep1.astype(bool))
        assert list(o0.rvec) == list(o1.rvec) and list(o0.tvec) == list(o1.tvec) and list(o0.R) == list(o1.R)
        assert np.array_equal(rng, st0)


def test_pnp_tracker_mirror_against_the_reference_class_fixture(gpu):
    """the PnpTracker mirror (libs/tracker/pnp_tracker.py of the package, dfvo_compute_pose_3d2d underneath) on the cases of
    tests/golden/pnp_tracker.npz -- written by the REFERENCE's own PnpTracker.compute_pose_3d2d over the oracle cv2
    (make_golden.py:golden_pnp_tracker): surviving keypoints and the global numpy RandomState bit for bit, pose (after the
    final inversion, computed on the host by the mirror) to 1e-12; 5 and 3 repeats, few points, four points, a coplanar object"""
    import os
    from golden.make_golden import PNP_CASES, pnp_case as fixture_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pnp_tracker.npz"))
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    trk_mod = importlib.import_module("df-vo_amd.libs.tracker")
    for tag, (seed, n, of, noise, it, cop) in PNP_CASES.items():
        c = fixture_case(seed, n, of, noise, cop)
        Kc = c["K"]
        tracker = trk_mod.PnpTracker(_cfg(), cam_mod.Intrinsics([Kc[0, 2], Kc[1, 2], Kc[0, 0], Kc[1, 1]]))
        np.random.seed(4869 + seed)
        res = tracker.compute_pose_3d2d(c["kp1"], c["kp2"], c["depth_1"], bool(it))
        assert np.array_equal(res["kp1"], g[tag + "_kp1"]) and np.array_equal(res["kp2"], g[tag + "_kp2"]), tag
        assert np.array_equal(np_state(), g[tag + "_rng_after"]), tag
        assert np.abs(res["pose"].pose - g[tag + "_pose"]).max() <= 1e-12, (tag, np.abs(res["pose"].pose - g[tag + "_pose"]).max())
