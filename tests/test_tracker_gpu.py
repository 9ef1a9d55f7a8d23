"""GPU parity of the device-side tracker stages (keypoint selection, compute_pose_2d2d, scale recovery)
against the oracle (oracle/tracker_np.py, pinned to the reference by tests/golden) on identical inputs
and an identical numpy RandomState.  Bit-exact: keypoints (values and order), inlier masks, R, t, the
RandomState after the call.  Within stated tolerance: GRIC scores (libm acos/sin/cos/log differ in the
last bits between numpy and the device library), the final least-squares scale (sklearn goes through
LAPACK gelsd)."""
import ctypes as C

import warnings

import numpy as np
import pytest

from golden.make_golden import kp_case, tracker_case
from oracle import tracker_np as T

pytestmark = pytest.mark.gpu


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


def push_rng(gpu, trk):
    s = np_state()
    gpu.check(gpu.lib().dfvo_tracker_set_rng_state(trk, gpu.as_ptr(s)))


def pull_rng(gpu, trk):
    s = np.zeros(625, np.uint32)
    gpu.check(gpu.lib().dfvo_tracker_get_rng_state(trk, gpu.as_ptr(s)))
    return s


def test_device_seed_equals_numpy_seed(gpu, trk):
    for seed in (4869, 0, 123456789):
        gpu.check(gpu.lib().dfvo_tracker_seed(trk, seed))
        np.random.seed(seed)
        assert np.array_equal(pull_rng(gpu, trk), np_state())


@pytest.mark.parametrize("h,w,seed,frac", [(192, 640, 11, 0.6), (376, 1241, 12, 0.35), (100, 130, 13, 0.02),
                                            (120, 160, 14, 0.004), (376, 1241, 15, 0.0005), (1280, 1920, 16, 0.2)])
def test_local_bestn_bit_exact(gpu, trk, h, w, seed, frac):
    diff, flow = kp_case(h, w, seed, frac)
    ref = T.local_bestN(flow, diff)
    kp1 = np.zeros((2000, 2))
    kp2 = np.zeros((2000, 2))
    n, good = C.c_int(), C.c_int()
    gpu.check(gpu.lib().dfvo_kp_local_bestn(trk, gpu.as_ptr(np.ascontiguousarray(flow)),
                                            gpu.as_ptr(np.ascontiguousarray(diff[..., 0])), h, w, 10, 10, 2000, 0.1,
                                            gpu.as_ptr(kp1), gpu.as_ptr(kp2), C.byref(n), C.byref(good)))
    print("local_bestN %dx%d: oracle good=%s n=%s | hip good=%d n=%d" % (
        h, w, ref["good_kp_found"], ref.get("kp1_best", np.zeros((1, 0, 2))).shape[1], good.value, n.value))
    assert bool(good.value) == bool(ref["good_kp_found"])
    if ref["good_kp_found"]:
        assert n.value == ref["kp1_best"].shape[1]
        assert np.array_equal(kp1[:n.value], ref["kp1_best"][0])
        assert np.array_equal(kp2[:n.value], ref["kp2_best"][0])


@pytest.mark.parametrize("h,w,seed,frac,thre", [(192, 640, 21, 0.6, 0.02), (376, 1241, 22, 0.35, 0.01), (100, 130, 23, 0.05, 0.02),
                                                 (376, 1241, 24, 0.35, 0.003)])
def test_local_bestn_flow_ratio_bit_exact(gpu, trk, h, w, seed, frac, thre):
    """cfg.kp_selection.local_bestN.score_method 'flow_ratio' (dfvo_kp_local_bestn_ex): values and order against the oracle
    (pinned to the reference's kp_selection.py fixture), including pixels with zero flow (ratio inf / nan: never selected)"""
    from golden.make_golden import kp_ratio_case
    diff, flow = kp_ratio_case(h, w, seed, frac)
    ref = T.local_bestN(flow, diff, thre=thre, score_method="flow_ratio")
    kp1, kp2 = np.zeros((2000, 2)), np.zeros((2000, 2))
    n, good = C.c_int(), C.c_int()
    gpu.check(gpu.lib().dfvo_kp_local_bestn_ex(trk, gpu.as_ptr(np.ascontiguousarray(flow)),
                                               gpu.as_ptr(np.ascontiguousarray(diff[..., 0])), h, w, 10, 10, 2000, thre, 1,
                                               gpu.as_ptr(kp1), gpu.as_ptr(kp2), C.byref(n), C.byref(good)))
    assert bool(good.value) == bool(ref["good_kp_found"])
    if ref["good_kp_found"]:
        assert n.value == ref["kp1_best"].shape[1]
        assert np.array_equal(kp1[:n.value], ref["kp1_best"][0])
        assert np.array_equal(kp2[:n.value], ref["kp2_best"][0])


@pytest.mark.parametrize("h,w,seed,levels", [(376, 1241, 41, 8), (376, 1241, 42, 64), (192, 640, 43, 3), (376, 1241, 44, 1000)])
def test_local_bestn_heavy_ties(gpu, trk, h, w, seed, levels):
    """quantised consistency maps: most candidates of a cell tie with the pivot, which drives the workgroup-parallel
    partition through its equal-to-pivot (self-pair) paths; order must still be numpy's scalar introselect order"""
    rng = np.random.Generator(np.random.PCG64(seed))
    diff = (np.floor(rng.random((h, w, 1)) * levels) / levels * 0.12).astype(np.float32)
    flow = (rng.standard_normal((2, h, w)) * 3).astype(np.float32)
    ref = T.local_bestN(flow, diff)
    kp1, kp2 = np.zeros((2000, 2)), np.zeros((2000, 2))
    n, good = C.c_int(), C.c_int()
    gpu.check(gpu.lib().dfvo_kp_local_bestn(trk, gpu.as_ptr(np.ascontiguousarray(flow)),
                                            gpu.as_ptr(np.ascontiguousarray(diff[..., 0])), h, w, 10, 10, 2000, 0.1,
                                            gpu.as_ptr(kp1), gpu.as_ptr(kp2), C.byref(n), C.byref(good)))
    assert bool(good.value) == bool(ref["good_kp_found"]) and ref["good_kp_found"]
    assert n.value == ref["kp1_best"].shape[1]
    assert np.array_equal(kp1[:n.value], ref["kp1_best"][0])
    assert np.array_equal(kp2[:n.value], ref["kp2_best"][0])


CASES = [(31, 2000, 0.3, 0.15), (32, 2000, 0.6, 0.3), (33, 600, 0.2, 0.1), (34, 2000, 0.97, 0.2), (35, 1234, 0.4, 0.2),
         (36, 9, 0.0, 0.1)]


@pytest.mark.parametrize("seed,n,of,noise", CASES)
def test_compute_pose_2d2d_and_scale(gpu, trk, seed, n, of, noise):
    lib = gpu.lib()
    c = tracker_case(seed, n, of, noise)
    K = c["K"]
    np.random.seed(4869 + seed)
    push_rng(gpu, trk)
    ref = T.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], K)
    cfg = gpu.Pose2d2dCfg(fx=K[0, 0], cx=K[0, 2], cy=K[1, 2], reproj_thre=0.2, repeat=5, max_iters=1000)
    KinvT, Kinv = np.linalg.inv(K.T), np.linalg.inv(K)
    for i in range(9):
        cfg.KinvT[i] = KinvT.flat[i]
        cfg.Kinv[i] = Kinv.flat[i]
    out = gpu.Pose2d2dOut()
    inl = np.zeros(max(n, 1), np.uint8)
    gpu.check(lib.dfvo_compute_pose_2d2d(trk, gpu.as_ptr(c["kp_ref"]), gpu.as_ptr(c["kp_cur"]), n, C.byref(cfg),
                                         C.byref(out), gpu.as_ptr(inl)))
    R = np.array(out.R[:]).reshape(3, 3)
    t = np.array(out.t[:]).reshape(3, 1)
    print("pose2d2d n=%d: oracle reps %s valid %s cheir %d | hip reps %s valid %s cheir %d" % (
        n, ref["rep_inliers"], ref["rep_valid"], ref["cheirality"], list(out.rep_inliers[:5]), list(out.rep_valid[:5]),
        out.cheirality))
    if n > 10:
        assert list(out.rep_inliers[:5]) == ref["rep_inliers"]
        assert [bool(v) for v in out.rep_valid[:5]] == ref["rep_valid"]
        assert abs(out.h_gric - ref["h_gric"]) <= 1e-9 * abs(ref["h_gric"])
        for a, b in zip(out.rep_gric[:5], ref["rep_gric"]):
            assert abs(a - b) <= 1e-9 * abs(b)
        assert out.cheirality == ref["cheirality"]
    assert np.array_equal(inl[:n] == 1, ref["inliers"])
    assert np.array_equal(R, ref["R"]) and np.array_equal(t, ref["t"])
    assert np.array_equal(pull_rng(gpu, trk), np_state()), "RandomState diverged after compute_pose_2d2d"
    if np.linalg.norm(ref["t"]) == 0:
        return
    # ---- scale recovery with the same pose
    pose = np.eye(4)
    pose[:3, :3] = ref["R"]
    pose[:3, 3:] = ref["t"]
    T21 = np.ascontiguousarray(np.linalg.inv(pose))
    diag = {}
    s_ref = T.find_scale_from_depth(c["kp_ref"], c["kp_cur"], T21, c["depth_cur"], K, diag=diag)
    scfg = gpu.ScaleCfg(cx=K[0, 2], cy=K[1, 2], fx=K[0, 0], fy=K[1, 1], min_samples=3, max_trials=100, stop_prob=0.99,
                        thre=0.1)
    scale = C.c_double()
    info = np.zeros(4, np.int32)
    h, w = c["depth_cur"].shape
    gpu.check(lib.dfvo_find_scale_from_depth(trk, gpu.as_ptr(c["kp_ref"]), gpu.as_ptr(c["kp_cur"]), n, gpu.as_ptr(T21),
                                             gpu.as_ptr(np.ascontiguousarray(c["depth_cur"])), h, w, C.byref(scfg),
                                             C.byref(scale), gpu.as_ptr(info)))
    print("scale: oracle %.15g (valid %s trials %s inliers %s) | hip %.15g info %s" % (
        s_ref, diag.get("n_valid"), diag.get("n_trials"), diag.get("n_inliers"), scale.value, info.tolist()))
    assert info[0] == diag["n_valid"]
    if s_ref == -1:
        assert scale.value == -1
    else:
        assert info[1] == diag["n_trials"] and info[2] == diag["n_inliers"]
        assert abs(scale.value - s_ref) <= 1e-12 * abs(s_ref)
    assert np.array_equal(pull_rng(gpu, trk), np_state()), "RandomState diverged after scale recovery"


def test_scale_recovery_depth_at_keypoints_entry(gpu, trk):
    """dfvo_find_scale_from_depth_at_kp (the depth map's values at the truncated kp2 pixels instead of the H x W map, RandomState in
    and out of the same call -- what the EssTracker mirror uses) returns what dfvo_find_scale_from_depth returns, bit for bit,
    keypoints outside the map / with non-finite coordinates / sharing a pixel included"""
    lib = gpu.lib()
    for seed, n in ((51, 600), (52, 2000), (53, 40)):
        c = tracker_case(seed, n, 0.1, 0.05)
        K = c["K"]
        pose = np.eye(4)
        pose[:3, :3] = c["R"]
        pose[:3, 3] = c["t"]
        T21 = np.ascontiguousarray(pose)
        kp1, kp2 = c["kp_ref"].copy(), c["kp_cur"].copy()
        h, w = c["depth_cur"].shape
        kp2[3] = [-0.5, 10.2]            # truncates to pixel (0, 10): inside
        kp2[5] = [-1.5, 10.2]            # outside
        kp2[7] = [w + 0.25, 3.0]         # outside
        kp2[9] = [np.nan, 5.0]
        kp2[11] = [12.0, np.inf]
        kp2[13] = kp2[12] + 0.25         # (most likely) the same pixel as its neighbour: the later one wins
        depth = np.ascontiguousarray(c["depth_cur"], dtype=np.float64)
        scfg = gpu.ScaleCfg(cx=K[0, 2], cy=K[1, 2], fx=K[0, 0], fy=K[1, 1], min_samples=3, max_trials=100, stop_prob=0.99, thre=0.1)
        np.random.seed(seed)
        push_rng(gpu, trk)
        s0, i0 = C.c_double(), np.zeros(4, np.int32)
        gpu.check(lib.dfvo_find_scale_from_depth(trk, gpu.as_ptr(kp1), gpu.as_ptr(kp2), n, gpu.as_ptr(T21), gpu.as_ptr(depth), h, w,
                                                 C.byref(scfg), C.byref(s0), gpu.as_ptr(i0)))
        rng0 = pull_rng(gpu, trk)
        tx, ty = np.trunc(kp2[:, 0]), np.trunc(kp2[:, 1])
        with np.errstate(invalid="ignore"):
            inside = (tx >= 0) & (tx < w) & (ty >= 0) & (ty < h)
        at_kp = np.full(n, -7.0)         # (values at dropped keypoints must not matter)
        at_kp[inside] = depth[ty[inside].astype(int), tx[inside].astype(int)]
        np.random.seed(seed)
        st = np.random.get_state()
        rng = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
        s1, i1 = C.c_double(), np.zeros(4, np.int32)
        gpu.check(lib.dfvo_find_scale_from_depth_at_kp(trk, gpu.as_ptr(kp1), gpu.as_ptr(kp2), n, gpu.as_ptr(T21), gpu.as_ptr(at_kp), h, w,
                                                       C.byref(scfg), gpu.as_ptr(rng), C.byref(s1), gpu.as_ptr(i1)))
        print("n=%d map entry %.15g %s | per-keypoint entry %.15g %s" % (n, s0.value, i0.tolist(), s1.value, i1.tolist()))
        assert s0.value == s1.value and np.array_equal(i0, i1) and i0[0] > 10
        assert np.array_equal(rng, rng0), "RandomState returned by the call differs from the tracker's"
        assert np.array_equal(pull_rng(gpu, trk), rng0)


def test_scale_recovery_small_populations(gpu, trk):
    """exercise sklearn's sample_without_replacement branches: permutation (n < 300) / tracking selection"""
    lib = gpu.lib()
    for seed, n in ((41, 40), (42, 150), (43, 299), (44, 301), (45, 12), (46, 8)):
        c = tracker_case(seed, n, 0.1, 0.05)
        K = c["K"]
        pose = np.eye(4)
        pose[:3, :3] = c["R"]
        pose[:3, 3] = c["t"]
        T21 = np.ascontiguousarray(pose)  # true relative pose ref -> cur
        np.random.seed(seed)
        push_rng(gpu, trk)
        diag = {}
        s_ref = T.find_scale_from_depth(c["kp_ref"], c["kp_cur"], T21, c["depth_cur"], K, diag=diag)
        scfg = gpu.ScaleCfg(cx=K[0, 2], cy=K[1, 2], fx=K[0, 0], fy=K[1, 1], min_samples=3, max_trials=100,
                            stop_prob=0.99, thre=0.1)
        scale = C.c_double()
        info = np.zeros(4, np.int32)
        h, w = c["depth_cur"].shape
        gpu.check(lib.dfvo_find_scale_from_depth(trk, gpu.as_ptr(c["kp_ref"]), gpu.as_ptr(c["kp_cur"]), n,
                                                 gpu.as_ptr(T21), gpu.as_ptr(np.ascontiguousarray(c["depth_cur"])), h,
                                                 w, C.byref(scfg), C.byref(scale), gpu.as_ptr(info)))
        print("n=%d oracle %.12g %s | hip %.12g %s" % (n, s_ref, {k: v for k, v in diag.items() if k != "ratios"},
                                                       scale.value, info.tolist()))
        assert info[0] == diag["n_valid"]
        if s_ref == -1:
            assert scale.value == -1
        else:
            assert info[1] == diag["n_trials"] and info[2] == diag["n_inliers"]
            assert abs(scale.value - s_ref) <= 1e-12 * abs(s_ref)
        assert np.array_equal(pull_rng(gpu, trk), np_state())


def test_scale_recovery_sklearn_versions(gpu, trk):
    """dfvo_set_sklearn_compat: a residual threshold so tight that consensus sets of ONE sample occur -- r2_score of one
    sample is nan in the installed scikit-learn (>= 0.22) and 1.0 / 0.0 in 0.20.3, the reference's pin.  Device vs the
    installed RANSACRegressor with the matching r2_score (oracle/sklearn_compat.py), both rules: trial count, inlier count,
    coefficient, RandomState.  (The two rules only part ways when a one-sample consensus set has an exactly zero residual
    -- score 1.0 instead of 0.0 / nan -- which none of 360 random cases reached; tests/test_oracle_sklearn_compat.py shows
    the difference on a crafted input.)"""
    import sklearn
    from oracle.sklearn_compat import r2_score_like
    lib = gpu.lib()
    results = {}
    try:
        for version in ("0.20.3", sklearn.__version__):
            gpu.set_sklearn_compat(version)
            for seed, n, thre in ((51, 60, 2e-3), (52, 200, 1e-3), (53, 400, 5e-4), (54, 30, 1e-3), (55, 20, 3e-3), (56, 14, 1e-3)):
                c = tracker_case(seed, n, 0.1, 0.05)
                K = c["K"]
                pose = np.eye(4)
                pose[:3, :3] = c["R"]
                pose[:3, 3] = c["t"]
                T21 = np.ascontiguousarray(pose)
                np.random.seed(seed)
                push_rng(gpu, trk)
                diag = {}
                raised = False
                with r2_score_like(version), warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    try:
                        s_ref = T.find_scale_from_depth(c["kp_ref"], c["kp_cur"], T21, c["depth_cur"], K, diag=diag, thre=thre)
                    except ValueError:  # "RANSAC could not find a valid consensus set"
                        raised = True
                scfg = gpu.ScaleCfg(cx=K[0, 2], cy=K[1, 2], fx=K[0, 0], fy=K[1, 1], min_samples=3, max_trials=100,
                                    stop_prob=0.99, thre=thre)
                scale = C.c_double()
                info = np.zeros(4, np.int32)
                h, w = c["depth_cur"].shape
                gpu.check(lib.dfvo_find_scale_from_depth(trk, gpu.as_ptr(c["kp_ref"]), gpu.as_ptr(c["kp_cur"]), n,
                                                         gpu.as_ptr(T21), gpu.as_ptr(np.ascontiguousarray(c["depth_cur"])),
                                                         h, w, C.byref(scfg), C.byref(scale), gpu.as_ptr(info)))
                print("sklearn %s seed %d: oracle %s %s | hip %.12g %s" % (
                    version, seed, "raised" if raised else "%.12g" % s_ref, {k: v for k, v in diag.items() if k != "ratios"},
                    scale.value, info.tolist()))
                assert np.array_equal(pull_rng(gpu, trk), np_state())
                if raised:
                    assert info[3] == -1
                    continue
                assert info[0] == diag["n_valid"]
                if s_ref == -1:
                    assert scale.value == -1
                else:
                    assert info[1] == diag["n_trials"] and info[2] == diag["n_inliers"]
                    assert abs(scale.value - s_ref) <= 1e-12 * abs(s_ref)
                results[(version, seed)] = (scale.value, int(info[1]), int(info[2]))
    finally:
        gpu.set_sklearn_compat(sklearn.__version__)
    one_sample = [k for k, v in results.items() if v[2] == 1]
    print("results with a one-sample consensus set:", one_sample)
    assert one_sample, "no case ended on a one-sample consensus set: the test does not reach the rule"


def test_ransac_regressor_one_sample_consensus_rule(gpu, trk):
    """dfvo_ransac_regressor on the crafted input of tests/test_oracle_sklearn_compat.py (zeros, isolated powers of two, two
    small values; y = 1): one-sample consensus sets with exactly zero and with non-zero residuals alternate, so the r2_score
    rule of scikit-learn 0.20.3 (1.0 / 0.0) and of >= 0.22 (nan) pick different final sets on half of the seeds.  The
    device follows dfvo_set_sklearn_compat, bit for bit: coefficient, inlier count, trial count, RandomState."""
    import sklearn
    from sklearn import linear_model
    from oracle.sklearn_compat import r2_score_like
    lib = gpu.lib()
    x = np.array([0.0] * 6 + [4.0, 64.0, 1024.0, 16384.0] + [0.5, 0.25])
    got = {}
    try:
        for version in ("0.20.3", sklearn.__version__):
            gpu.set_sklearn_compat(version)
            for seed in range(30):
                np.random.seed(seed)
                push_rng(gpu, trk)
                r = linear_model.RANSACRegressor(estimator=linear_model.LinearRegression(fit_intercept=False), min_samples=3,
                                                 max_trials=40, stop_probability=0.99, residual_threshold=0.3)
                want = None
                with r2_score_like(version), warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    try:
                        r.fit(x.reshape(-1, 1), np.ones((x.shape[0], 1)))
                        want = (float(r.estimator_.coef_[0, 0]), int(r.n_trials_), int(r.inlier_mask_.sum()))
                    except ValueError:
                        pass
                scfg = gpu.ScaleCfg(cx=0, cy=0, fx=1, fy=1, min_samples=3, max_trials=40, stop_prob=0.99, thre=0.3)
                coef = C.c_double()
                info = np.zeros(4, np.int32)
                gpu.check(lib.dfvo_ransac_regressor(trk, gpu.as_ptr(x), None, x.shape[0], C.byref(scfg), C.byref(coef),
                                                    gpu.as_ptr(info)))
                assert np.array_equal(pull_rng(gpu, trk), np_state()), (version, seed)
                if want is None:
                    assert info[3] == -1, (version, seed)
                else:
                    assert (coef.value, int(info[1]), int(info[2])) == want, (version, seed, coef.value, info.tolist(), want)
                got[(version, seed)] = want
    finally:
        gpu.set_sklearn_compat(sklearn.__version__)
    differ = [s for s in range(30) if got[("0.20.3", s)] != got[(sklearn.__version__, s)]]
    print("seeds on which the two scikit-learn rules end on different consensus sets:", differ)
    major, minor = (int(v) for v in sklearn.__version__.split(".")[:2])
    assert differ or not (major > 0 or minor >= 22)


def _pose2d2d(gpu, trk, kp_ref, kp_cur, K, **kw):
    n = kp_ref.shape[0]
    cfg = gpu.Pose2d2dCfg(fx=K[0, 0], cx=K[0, 2], cy=K[1, 2], reproj_thre=0.2, repeat=5, max_iters=1000, **kw)
    KinvT, Kinv = np.linalg.inv(K.T), np.linalg.inv(K)
    for i in range(9):
        cfg.KinvT[i] = KinvT.flat[i]
        cfg.Kinv[i] = Kinv.flat[i]
    out = gpu.Pose2d2dOut()
    inl = np.zeros(max(n, 1), np.uint8)
    gpu.check(gpu.lib().dfvo_compute_pose_2d2d(trk, gpu.as_ptr(np.ascontiguousarray(kp_ref)),
                                               gpu.as_ptr(np.ascontiguousarray(kp_cur)), n, C.byref(cfg), C.byref(out),
                                               gpu.as_ptr(inl)))
    return out, inl[:n] == 1


@pytest.mark.parametrize("tag", list("abcd"))
def test_compute_pose_2d2d_flow_validity(gpu, trk, tag):
    """e_tracker.validity.method 'flow' (ablation_model_sel_flow.yml) against the reference fixture and the oracle:
    pose, inlier mask, per-repeat counts and the RandomState afterwards (a closed gate draws nothing)"""
    import os
    from test_oracle_tracker import flow_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e_tracker_flow.npz"))
    seed, kp_ref, kp_cur, K = flow_case(g, tag)
    np.random.seed(4869 + seed)
    push_rng(gpu, trk)
    ref = T.compute_pose_2d2d(kp_ref, kp_cur, K, validity="flow", validity_thre=5)
    out, inl = _pose2d2d(gpu, trk, kp_ref, kp_cur, K, validity_method=1, validity_thre=5.0)
    R = np.array(out.R[:]).reshape(3, 3)
    t = np.array(out.t[:]).reshape(3, 1)
    print("flow validity %s: avg flow %.3f | oracle reps %s cheir %s valid %s | hip reps %s cheir %s valid %s" % (
        tag, ref["avg_flow"], ref["rep_inliers"], ref["rep_cheirality"], ref["rep_valid"], list(out.rep_inliers[:5]),
        [int(v) for v in out.rep_gric[:5]], list(out.rep_valid[:5])))
    assert abs(out.h_gric - ref["avg_flow"]) == 0  # numpy's pairwise mean, bit for bit
    if ref["rep_inliers"]:
        assert list(out.rep_inliers[:5]) == ref["rep_inliers"]
        assert [int(v) for v in out.rep_gric[:5]] == ref["rep_cheirality"]
        assert [bool(v) for v in out.rep_valid[:5]] == ref["rep_valid"]
    else:
        assert out.num_valid == 0 and out.best_inlier_cnt == 0
    pose = g[tag + "_pose"]
    assert np.array_equal(inl, ref["inliers"]) and np.array_equal(inl, g[tag + "_inliers"])
    assert np.array_equal(R, ref["R"]) and np.array_equal(t, ref["t"])
    assert np.array_equal(R, pose[:3, :3]) and np.array_equal(t, pose[:3, 3:])
    assert np.array_equal(pull_rng(gpu, trk), np_state()), "RandomState diverged"
    assert np.array_equal(pull_rng(gpu, trk), g[tag + "_rng_after"])


@pytest.mark.parametrize("tag", list("abpd"))
def test_compute_pose_2d2d_homo_ratio_and_abs_diff_scale(gpu, trk, tag):
    """e_tracker.validity.method 'homo_ratio' and scale_recovery.ransac.method 'abs_diff' (E_tracker.py:186-194,243-250,
    631-635) against the reference fixture and the oracle: pose, inlier mask, per-repeat counts / ratios bit-exact, the
    least-squares scale to 1e-12 (sklearn goes through LAPACK), RandomState afterwards identical"""
    import os
    from golden.make_golden import variant_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e_tracker_variants.npz"))
    c = variant_case(tag)
    kp_ref, kp_cur, K = c["kp_ref"], c["kp_cur"], c["K"]
    np.random.seed(4869 + c["seed"])
    push_rng(gpu, trk)
    ref = T.compute_pose_2d2d(kp_ref, kp_cur, K, validity="homo_ratio", validity_thre=0.4)
    out, inl = _pose2d2d(gpu, trk, kp_ref, kp_cur, K, validity_method=2, validity_thre=0.4)
    R = np.array(out.R[:]).reshape(3, 3)
    t = np.array(out.t[:]).reshape(3, 1)
    print("homo_ratio %s: H inliers %d | oracle reps %s ratio %s valid %s | hip reps %s ratio %s valid %s" % (
        tag, ref["h_inliers"], ref["rep_inliers"], np.round(ref["rep_ratio"], 4), ref["rep_valid"],
        list(out.rep_inliers[:5]), np.round(out.rep_gric[:5], 4), list(out.rep_valid[:5])))
    assert out.h_gric == ref["h_inliers"]
    assert list(out.rep_inliers[:5]) == ref["rep_inliers"]
    assert list(out.rep_gric[:5]) == ref["rep_ratio"]
    assert [bool(v) for v in out.rep_valid[:5]] == ref["rep_valid"]
    pose = g[tag + "_pose"]
    assert np.array_equal(inl, ref["inliers"]) and np.array_equal(inl, g[tag + "_inliers"])
    assert np.array_equal(R, ref["R"]) and np.array_equal(t, ref["t"])
    assert np.array_equal(R, pose[:3, :3]) and np.array_equal(t, pose[:3, 3:])
    assert np.array_equal(pull_rng(gpu, trk), np_state()), "RandomState diverged"
    if np.linalg.norm(t) != 0:
        T21 = np.linalg.inv(pose)
        diag = {}
        s_ref = T.find_scale_from_depth(kp_ref, kp_cur, T21, c["depth_cur"], K, diag=diag, method="abs_diff")
        h, w = c["depth_cur"].shape
        scfg = gpu.ScaleCfg(cx=K[0, 2], cy=K[1, 2], fx=K[0, 0], fy=K[1, 1], min_samples=3, max_trials=100, stop_prob=0.99,
                            thre=0.1, method=1)
        scale = C.c_double()
        info = np.zeros(4, np.int32)
        gpu.check(gpu.lib().dfvo_find_scale_from_depth(trk, gpu.as_ptr(kp_ref), gpu.as_ptr(kp_cur), kp_ref.shape[0],
                                                       gpu.as_ptr(np.ascontiguousarray(T21)), gpu.as_ptr(c["depth_cur"]), h, w,
                                                       C.byref(scfg), C.byref(scale), gpu.as_ptr(info)))
        print("abs_diff scale: oracle %.15g (valid %s trials %s inliers %s) | hip %.15g info %s" % (
            s_ref, diag.get("n_valid"), diag.get("n_trials"), diag.get("n_inliers"), scale.value, info.tolist()))
        assert info[0] == diag["n_valid"] and info[1] == diag["n_trials"] and info[2] == diag["n_inliers"]
        assert abs(scale.value - s_ref) <= 1e-12 * abs(s_ref)
        assert abs(scale.value - float(g[tag + "_scale"])) <= 1e-12 * abs(s_ref)
        assert np.array_equal(pull_rng(gpu, trk), np_state()), "RandomState diverged after scale recovery"
    assert np.array_equal(pull_rng(gpu, trk), g[tag + "_rng_after"])


@pytest.mark.parametrize("tag", list("abc"))
def test_sampled_kp_bit_exact(gpu, trk, tag):
    """sampled_kp (ablation_correspondences_uniform.yml) against the reference fixture"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampled_kp.npz"))
    spec = g[tag + "_spec"]
    h, w, seed, nkp = [int(v) for v in spec[:4]]
    crop = [[float(spec[4]), float(spec[5])], [float(spec[6]), float(spec[7])]]
    _, flow = kp_case(h, w, seed, 0.5)
    idx = np.ascontiguousarray(T.generate_kp_samples(h, w, crop, nkp), np.int32)
    y0, y1, x0, x1 = int(h * crop[0][0]), int(h * crop[0][1]), int(w * crop[1][0]), int(w * crop[1][1])
    kp1, kp2 = np.zeros((nkp, 2)), np.zeros((nkp, 2))
    gpu.check(gpu.lib().dfvo_kp_sampled(trk, gpu.as_ptr(np.ascontiguousarray(flow)), h, w, y0, y1, x0, x1, gpu.as_ptr(idx),
                                        nkp, gpu.as_ptr(kp1), gpu.as_ptr(kp2)))
    assert np.array_equal(kp1, g[tag + "_kp1"][0]) and np.array_equal(kp2, g[tag + "_kp2"][0])


@pytest.mark.parametrize("tag", list("abc"))
def test_rigid_flow_kp_and_iterative_scale(gpu, trk, tag):
    """SURVEY 8f rank 1 (ablation_scale_iterative.yml): RigidFlow layer + opt_rigid_flow_kp + scale_recovery_iterative
    through the mirror classes, against the fixtures produced by the reference's own code on CPU torch.  Float32
    rigid-flow distance map: bit-exact (the fused multiply-add order of the torch-CPU GEMM that produced the fixture); keypoints: bit-exact values
    and order; iterative scale: 1e-12 relative; RandomState afterwards identical."""
    import importlib
    import os
    import zlib
    from types import SimpleNamespace as NS
    from golden.make_golden import RIGID_CASES, rigid_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rigid_flow_kp.npz"))
    h, w, seed, score = RIGID_CASES[tag]
    c = rigid_case(h, w, seed)
    K = c["K"]
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    E_mod = importlib.import_module("df-vo_amd.libs.tracker.E_tracker")
    ctx = importlib.import_module("df-vo_amd.libs.tracker._ctx")
    rfk = NS(enable=True, num_bestN=2000, num_row=10, num_col=10, score_method=score, rigid_flow_thre=5, optical_flow_thre=0.1)
    cfg = NS(kp_selection=NS(rigid_flow_kp=rfk),
             e_tracker=NS(ransac=NS(reproj_thre=0.2, repeat=5), validity=NS(method="GRIC", thre=None), kp_src="kp_best",
                          iterative_kp=NS(enable=False, kp_src="kp_depth", score_method=score)),
             scale_recovery=NS(method="iterative", kp_src="kp_depth",
                               iterative_kp=NS(enable=False, kp_src="kp_depth", score_method=score),
                               ransac=NS(method="depth_ratio", min_samples=3, max_trials=100, stop_prob=0.99, thre=0.1)),
             image=NS(height=h, width=w))
    cam = cam_mod.Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
    et = E_mod.EssTracker(cfg, cam, None)
    ref = {"flow": c["flow"], "flow_diff": c["diff"][..., None], "raw_depth": c["raw_depth"],
           "rigid_flow_pose": cam_mod.SE3(c["T_ref_to_cur"].copy())}
    cur = {"depth": c["depth_cur"]}
    res = et.kp_selection_good_depth(cur, ref, score)
    m = np.ascontiguousarray(res["rigid_flow_mask"], np.float32)
    want = T.kp_selection_good_depth(c["flow"], c["diff"][..., None], c["raw_depth"], c["T_ref_to_cur"], K, score)
    nbad = int((m != want["rigid_flow_mask"]).sum())
    print("rigid-flow distance map %s: %d / %d differ from the oracle, max |diff| %.3g" % (
        tag, nbad, m.size, float(np.abs(m - want["rigid_flow_mask"]).max())))
    assert nbad == 0
    assert zlib.crc32(m.tobytes()) == int(g[tag + "_mask_crc"])
    for k in ("kp1_depth", "kp2_depth", "kp1_depth_uniform", "kp2_depth_uniform"):
        assert np.array_equal(res[k], g[tag + "_" + k]), (tag, k)
    # scale_recovery_iterative from prev_scale = 0 (the numpy stream runs through the device-resident RandomState)
    E_pose = cam_mod.SE3(np.linalg.inv(c["T_ref_to_cur"]))
    E_pose.t = E_pose.t / np.linalg.norm(E_pose.t)
    np.random.seed(4869 + seed)
    et.prev_scale = 0
    out = et.scale_recovery(cur, ref, E_pose, False)
    assert abs(out["scale"] - float(g[tag + "_iter_scale"])) <= 1e-12 * abs(out["scale"])
    assert np.array_equal(out["cur_kp_depth"], g[tag + "_iter_cur_kp"]) and np.array_equal(out["ref_kp_depth"], g[tag + "_iter_ref_kp"])
    st = np.random.get_state()
    assert np.array_equal(np.r_[st[1].astype(np.uint32), np.uint32(st[2])], g[tag + "_rng_after"])


@pytest.mark.parametrize("tag", list("abcd"))
def test_bestn_flow_kp_bit_exact(gpu, trk, tag):
    """bestN_flow_kp (ablation_correspondences_best_n.yml): whole-image selection in numpy's introselect order, against
    the fixture produced by the reference's own kp_selection.py and against the oracle"""
    import os
    from golden.make_golden import BESTN_CASES, bestn_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bestN.npz"))
    h, w, seed, frac, N, hard = BESTN_CASES[tag]
    diff, flow = bestn_case(h, w, seed, frac, hard)
    kp1, kp2 = np.zeros((N, 2)), np.zeros((N, 2))
    n = C.c_int()
    gpu.check(gpu.lib().dfvo_kp_bestn(trk, gpu.as_ptr(np.ascontiguousarray(flow)),
                                      gpu.as_ptr(np.ascontiguousarray(diff.reshape(h, w))), h, w, N, gpu.as_ptr(kp1),
                                      gpu.as_ptr(kp2), C.byref(n)))
    assert n.value == N
    assert np.array_equal(kp1, g[tag + "_kp1"][0]) and np.array_equal(kp2, g[tag + "_kp2"][0])
    o1, o2 = T.bestN_flow_kp(flow, diff, N)
    assert np.array_equal(kp1, o1[0]) and np.array_equal(kp2, o2[0])


def test_bestn_large_image_vs_oracle(gpu, trk):
    """1280 x 1920 (BASELINE config 5 size): 2.4 M candidates through the workgroup-parallel global-memory partition"""
    h, w, N = 1280, 1920, 2000
    diff, flow = kp_case(h, w, 81, 0.5)
    kp1, kp2 = np.zeros((N, 2)), np.zeros((N, 2))
    n = C.c_int()
    gpu.check(gpu.lib().dfvo_kp_bestn(trk, gpu.as_ptr(np.ascontiguousarray(flow)),
                                      gpu.as_ptr(np.ascontiguousarray(diff.reshape(h, w))), h, w, N, gpu.as_ptr(kp1),
                                      gpu.as_ptr(kp2), C.byref(n)))
    o1, o2 = T.bestN_flow_kp(flow, diff, N)
    assert n.value == N and np.array_equal(kp1, o1[0]) and np.array_equal(kp2, o2[0])
    # too few candidates: numpy raises "kth out of bounds"; the C entry reports zero keypoints
    gpu.check(gpu.lib().dfvo_kp_bestn(trk, gpu.as_ptr(np.ascontiguousarray(flow[:, :10, :10])),
                                      gpu.as_ptr(np.ascontiguousarray(diff[:10, :10, 0])), 10, 10, 100, gpu.as_ptr(kp1),
                                      gpu.as_ptr(kp2), C.byref(n)))
    assert n.value == 0
