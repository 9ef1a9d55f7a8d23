"""CPU: the oracle's host-side tracker restatement (oracle/tracker_np.py + oracle/cv3_*.c + np_select.c)
against fixtures produced by the reference's own kp_selection.py / gric.py / E_tracker.py
(tests/golden/make_golden.py), plus real numpy / scikit-learn where those ARE the third-party reference."""
import os
import subprocess
import sys

import numpy as np

from golden.make_golden import kp_case, tracker_case
from oracle import tracker_np as T

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIMD_OFF = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"


def test_argpartition_restatements_match_scalar_numpy():
    """pure-python and C restatements vs numpy itself with its SIMD sort dispatch disabled"""
    code = r"""
import numpy as np, sys
rng = np.random.default_rng(3)
out = []
for trial in range(60):
    n = int(rng.integers(1, 5000)); v = rng.random(n).astype(np.float32)
    if trial % 3 == 0: v = np.round(v * 50) / 50
    k = min(20, n)
    out.append(np.argpartition(v, k - 1)[:k])
np.save(sys.argv[1], np.concatenate(out))
"""
    path = "/tmp/_np_scalar_order.npy"
    env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=SIMD_OFF)
    subprocess.check_call([sys.executable, "-c", code, path], env=env)
    want = np.load(path)
    rng = np.random.default_rng(3)
    got_c, got_py = [], []
    for trial in range(60):
        n = int(rng.integers(1, 5000))
        v = rng.random(n).astype(np.float32)
        if trial % 3 == 0:
            v = np.round(v * 50) / 50
        k = min(20, n)
        got_c.append(T.argpartition_c(v, k - 1)[:k])
        if n < 1500:
            assert np.array_equal(T.argpartition_scalar(v, k - 1)[:k], got_c[-1])
    assert np.array_equal(np.concatenate(got_c), want)


def test_local_bestN_matches_reference_fixture():
    g = np.load(os.path.join(G, "local_bestN.npz"))
    for tag in "abcd":
        h, w, seed, frac = g[tag + "_spec"]
        diff, flow = kp_case(h, w, seed, frac)
        res = T.local_bestN(flow, diff)
        assert bool(res["good_kp_found"]) == bool(g[tag + "_good"])
        if res["good_kp_found"]:
            assert np.array_equal(res["kp1_best"], g[tag + "_kp1"]), tag
            assert np.array_equal(res["kp2_best"], g[tag + "_kp2"]), tag


def test_local_bestN_flow_ratio_matches_reference_fixture():
    from golden.make_golden import kp_ratio_case
    g = np.load(os.path.join(G, "local_bestN.npz"))
    n_good = 0
    for tag in ("ra", "rb", "rc"):
        h, w, seed, frac, thre = g[tag + "_spec"]
        diff, flow = kp_ratio_case(h, w, seed, frac)
        res = T.local_bestN(flow, diff, thre=float(thre), score_method="flow_ratio")
        assert bool(res["good_kp_found"]) == bool(g[tag + "_good"])
        if res["good_kp_found"]:
            n_good += 1
            assert np.array_equal(res["kp1_best"], g[tag + "_kp1"]), tag
            assert np.array_equal(res["kp2_best"], g[tag + "_kp2"]), tag
    assert n_good >= 2


def test_gric_matches_reference_fixture():
    g = np.load(os.path.join(G, "gric.npz"))
    f = T.fundamental_residual(g["F"], g["kp1"], g["kp2"])
    h = T.homography_residual(g["H"], g["kp1"], g["kp2"])
    assert np.allclose(f, g["f_res"], rtol=1e-12, atol=0)
    assert np.allclose(h, g["h_res"], rtol=1e-9, atol=1e-12)
    assert abs(T.calc_GRIC(f, 0.8, len(f), "EMat") - float(g["f_gric"])) < 1e-8
    assert abs(T.calc_GRIC(h, 0.8, len(h), "HMat") - float(g["h_gric"])) < 1e-8


def test_e_tracker_matches_reference_fixture():
    g = np.load(os.path.join(G, "e_tracker.npz"))
    for tag in "abcd":
        seed, n, of, noise = g[tag + "_spec"]
        c = tracker_case(int(seed), int(n), float(of), float(noise))
        np.random.seed(4869 + int(seed))
        res = T.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], c["K"])
        pose = g[tag + "_pose"]
        assert np.array_equal(res["inliers"], g[tag + "_inliers"]), tag
        assert np.array_equal(res["R"], pose[:3, :3]) and np.array_equal(res["t"], pose[:3, 3:]), tag
        if np.linalg.norm(res["t"]) != 0:
            T21 = np.linalg.inv(pose)
            scale = T.find_scale_from_depth(c["kp_ref"], c["kp_cur"], T21, c["depth_cur"], c["K"])
            assert abs(scale - float(g[tag + "_scale"])) <= 1e-12 * abs(scale), tag
        st = np.random.get_state()
        assert np.array_equal(np.r_[st[1].astype(np.uint32), np.uint32(st[2])], g[tag + "_rng_after"]), tag


def test_e_tracker_homo_ratio_and_abs_diff_match_reference_fixture():
    """e_tracker.validity.method 'homo_ratio' + scale_recovery.ransac.method 'abs_diff' (E_tracker.py:186-194,243-250,
    631-635): poses, masks, scale and the RandomState afterwards; case p (planar scene) is rejected by the ratio"""
    from golden.make_golden import variant_case
    g = np.load(os.path.join(G, "e_tracker_variants.npz"))
    rejected = 0
    for tag in "abpd":
        c = variant_case(tag)
        np.random.seed(4869 + c["seed"])
        res = T.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], c["K"], validity="homo_ratio", validity_thre=0.4)
        pose = g[tag + "_pose"]
        assert np.array_equal(res["inliers"], g[tag + "_inliers"]), tag
        assert np.array_equal(res["R"], pose[:3, :3]) and np.array_equal(res["t"], pose[:3, 3:]), tag
        if np.linalg.norm(res["t"]) != 0:
            scale = T.find_scale_from_depth(c["kp_ref"], c["kp_cur"], np.linalg.inv(pose), c["depth_cur"], c["K"],
                                            method="abs_diff")
            assert abs(scale - float(g[tag + "_scale"])) <= 1e-12 * abs(scale), tag
        else:
            rejected += 1
            assert float(g[tag + "_scale"]) == -2.0
        st = np.random.get_state()
        assert np.array_equal(np.r_[st[1].astype(np.uint32), np.uint32(st[2])], g[tag + "_rng_after"]), tag
    assert rejected == 1


def flow_case(g, tag):
    """inputs of the e_tracker_flow fixtures: tracker_case with the displacement scaled by `shrink`"""
    seed, n, of, noise, shrink = g[tag + "_spec"]
    c = tracker_case(int(seed), int(n), float(of), float(noise))
    kp_cur = c["kp_ref"] + (c["kp_cur"] - c["kp_ref"]) * float(shrink)
    return int(seed), c["kp_ref"], kp_cur, c["K"]


def test_e_tracker_flow_validity_matches_reference_fixture():
    """e_tracker.validity.method 'flow' (ablation_model_sel_flow.yml): poses, masks and the RandomState afterwards"""
    g = np.load(os.path.join(G, "e_tracker_flow.npz"))
    for tag in "abcd":
        seed, kp_ref, kp_cur, K = flow_case(g, tag)
        np.random.seed(4869 + seed)
        res = T.compute_pose_2d2d(kp_ref, kp_cur, K, validity="flow", validity_thre=5)
        pose = g[tag + "_pose"]
        assert np.array_equal(res["inliers"], g[tag + "_inliers"]), tag
        assert np.array_equal(res["R"], pose[:3, :3]) and np.array_equal(res["t"], pose[:3, 3:]), tag
        st = np.random.get_state()
        assert np.array_equal(np.r_[st[1].astype(np.uint32), np.uint32(st[2])], g[tag + "_rng_after"]), tag


def test_sampled_kp_matches_reference_fixture():
    """sampled_kp + generate_kp_samples (ablation_correspondences_uniform.yml)"""
    g = np.load(os.path.join(G, "sampled_kp.npz"))
    for tag in "abc":
        h, w, seed, nkp = [int(v) for v in g[tag + "_spec"][:4]]
        crop = [[float(g[tag + "_spec"][4]), float(g[tag + "_spec"][5])], [float(g[tag + "_spec"][6]), float(g[tag + "_spec"][7])]]
        _, flow = kp_case(h, w, seed, 0.5)
        idx = T.generate_kp_samples(h, w, crop, nkp)
        assert np.array_equal(idx, g[tag + "_idx"]), tag
        kp1, kp2 = T.sampled_kp(flow, idx, crop)
        assert np.array_equal(kp1, g[tag + "_kp1"]) and np.array_equal(kp2, g[tag + "_kp2"]), tag


def test_rigid_flow_kp_and_iterative_scale_match_reference_fixture():
    """RigidFlow layer + opt_rigid_flow_kp + scale_recovery_iterative (ablation_scale_iterative.yml)
    against fixtures produced by the reference's own E_tracker.py / kp_selection.py / geometry layers on CPU torch"""
    import zlib
    from golden.make_golden import RIGID_CASES, rigid_case
    g = np.load(os.path.join(G, "rigid_flow_kp.npz"))
    for tag, (h, w, seed, score) in RIGID_CASES.items():
        c = rigid_case(h, w, seed)
        res = T.kp_selection_good_depth(c["flow"], c["diff"][..., None], c["raw_depth"], c["T_ref_to_cur"], c["K"], score)
        for k in ("kp1_depth", "kp2_depth", "kp1_depth_uniform", "kp2_depth_uniform"):
            assert np.array_equal(res[k], g[tag + "_" + k]), (tag, k)
        m = np.ascontiguousarray(res["rigid_flow_mask"], np.float32)
        assert np.array_equal(m[:: max(1, h // 8)], g[tag + "_mask_rows"])
        assert zlib.crc32(m.tobytes()) == int(g[tag + "_mask_crc"])
        E_pose = np.linalg.inv(c["T_ref_to_cur"])
        E_pose[:3, 3] /= np.linalg.norm(E_pose[:3, 3])
        np.random.seed(4869 + seed)
        it = T.scale_recovery_iterative(c["flow"], c["diff"][..., None], c["raw_depth"], c["depth_cur"], E_pose, c["K"], 0,
                                        score)
        assert abs(it["scale"] - float(g[tag + "_iter_scale"])) <= 1e-12 * abs(it["scale"]), tag
        assert np.array_equal(it["cur_kp"], g[tag + "_iter_cur_kp"]) and np.array_equal(it["ref_kp"], g[tag + "_iter_ref_kp"])
        st = np.random.get_state()
        assert np.array_equal(np.r_[st[1].astype(np.uint32), np.uint32(st[2])], g[tag + "_rng_after"]), tag


def test_bestN_flow_kp_matches_reference_fixture():
    """bestN_flow_kp (ablation_correspondences_best_n.yml): whole-image argpartition order, incl. ties and NaNs"""
    from golden.make_golden import BESTN_CASES, bestn_case
    g = np.load(os.path.join(G, "bestN.npz"))
    for tag, (h, w, seed, frac, N, hard) in BESTN_CASES.items():
        diff, flow = bestn_case(h, w, seed, frac, hard)
        kp1, kp2 = T.bestN_flow_kp(flow, diff, N)
        assert np.array_equal(kp1, g[tag + "_kp1"]) and np.array_equal(kp2, g[tag + "_kp2"]), tag


def test_compute_pose_3d2d_matches_reference_fixture():
    """oracle/tracker_np.compute_pose_3d2d (the expected value of the device's PnP path) against the reference's own
    PnpTracker.compute_pose_3d2d run over the oracle cv2 (tests/golden/pnp_tracker.npz): surviving keypoints, pose after the
    final inversion and the RandomState after the shuffles bit for bit -- 5 repeats (is_iterative) and 3, few points, four
    points (no solve: identity, the shuffles still drawn), a coplanar object"""
    from golden.make_golden import PNP_CASES, pnp_case
    g = np.load(os.path.join(G, "pnp_tracker.npz"))
    for tag, (seed, n, of, noise, it, cop) in PNP_CASES.items():
        c = pnp_case(seed, n, of, noise, cop)
        np.random.seed(4869 + seed)
        res = T.compute_pose_3d2d(c["kp1"], c["kp2"], c["depth_1"], c["K"], min_depth=0.0, max_depth=50.0,
                                  repeat=5 if it else 3, iters=100, reproj_thre=1.0)
        assert np.array_equal(res["kp1"], g[tag + "_kp1"]) and np.array_equal(res["kp2"], g[tag + "_kp2"]), tag
        assert np.array_equal(res["pose"], g[tag + "_pose"]), (tag, np.abs(res["pose"] - g[tag + "_pose"]).max())
        st = np.random.get_state()
        assert np.array_equal(np.r_[st[1].astype(np.uint32), np.uint32(st[2])], g[tag + "_rng_after"]), tag
