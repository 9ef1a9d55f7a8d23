"""CPU: the oracle against a SECOND installation of the third-party libraries the reference calls on its hot path -- older
releases, closer to the reference's pins (envs/requirement.yml: numpy 1.16.2, scikit-learn 0.20.3, Pillow 6.0) than the test
interpreter's numpy 2.x / scikit-learn 1.7 / Pillow 12.  This image carries one under /opt/conda (python 3.9: numpy 1.26.4 --
the last line before np.argpartition got SIMD kernels --, scikit-learn 0.24.2 -- still `RANSACRegressor(base_estimator=)` as
the reference writes it --, Pillow 8.4.0); DFVO_SECOND_PYTHON names another.  Skips where there is none (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import tracker_np as T
from oracle.pil_resample import resize_lanczos_u8

HERE = os.path.dirname(os.path.abspath(__file__))


def _second_python():
    for cand in (os.environ.get("DFVO_SECOND_PYTHON"), "/opt/conda/bin/python3.9"):
        if cand and os.path.exists(cand):
            try:
                r = subprocess.run([cand, "-W", "ignore", "-c", "import numpy, sklearn, PIL; print(numpy.__version__, sklearn.__version__, PIL.__version__)"],
                                   capture_output=True, text=True, timeout=120)
            except (OSError, subprocess.TimeoutExpired):
                continue
            if r.returncode == 0:
                return cand, r.stdout.split()
    return None, None


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    py, vers = _second_python()
    if py is None:
        pytest.skip("no second python environment with numpy / scikit-learn / Pillow (set DFVO_SECOND_PYTHON)")
    d = tmp_path_factory.mktemp("second_env")
    rng = np.random.default_rng(20260925)
    inp = {}
    ks = []
    for t in range(120):  # score vectors of a keypoint cell: plain, many ties, NaNs, sorted, reversed
        n = int(rng.integers(1, 6000))
        v = rng.random(n).astype(np.float32)
        if t % 4 == 0:
            v = np.round(v * 50) / 50
        if t % 7 == 0:
            v[rng.integers(0, n, max(1, n // 50))] = np.nan
        if t % 9 == 0:
            v = np.sort(v)
        if t % 11 == 0:
            v = np.sort(v)[::-1].copy()
        inp["ap_v%d" % t] = v
        ks.append(int(min(n, rng.integers(1, 40))))
    inp["ap_k"] = np.array(ks)
    seeds, thres = [], []
    for i in range(40):  # depth ratios: a consensus cluster + outliers; few points; a one-sample consensus set (crafted)
        if i % 8 == 7:
            ratio = np.array([1.0, 5.0, 9.0, 14.0, 20.0, 27.0, 35.0, 44.0, 54.0, 65.0, 77.0, 90.0])
        else:
            n_in, n_out = int(rng.integers(8, 400)), int(rng.integers(0, 200))
            ratio = np.concatenate([rng.uniform(0.5, 3) + rng.normal(0, 0.03, n_in), rng.uniform(0.1, 6, n_out)])
            rng.shuffle(ratio)
        inp["rs_ratio%d" % i] = ratio
        seeds.append(int(rng.integers(0, 2 ** 31)))
        thres.append(float([0.1, 0.05, 0.3][i % 3]))
    inp["rs_n"], inp["rs_seed"], inp["rs_thre"] = np.array(40), np.array(seeds), np.array(thres)
    sizes = [((376, 1241), (640, 192)), ((370, 1226), (640, 192)), ((960, 1280), (640, 256)), ((64, 96), (640, 192)), ((192, 640), (640, 192)), ((200, 300), (96, 64))]
    for i, (hw, wh) in enumerate(sizes):
        inp["pil_img%d" % i] = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    inp["pil_n"], inp["pil_size"] = np.array(len(sizes)), np.array([s[1] for s in sizes])
    src, dst = str(d / "in.npz"), str(d / "out.npz")
    np.savez(src, **inp)
    r = subprocess.run([py, "-W", "ignore", os.path.join(HERE, "second_env_probe.py"), src, dst], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = np.load(dst)
    print("second environment:", py, "numpy %s scikit-learn %s Pillow %s" % tuple(out["versions"]))
    return inp, out


def test_argpartition_restatement_is_the_order_of_numpy_1x(probe):
    """numpy 1.x has ONE np.argpartition (scalar introselect, npysort/selection.c.src -- the code the reference's pinned
    1.16.2 runs); numpy 2.x dispatches float32 keys to AVX-512 / AVX2 quickselect kernels whose first k indices come in
    another order.  The oracle's restatement (oracle/np_select.c, what the device reproduces) must be the 1.x order."""
    inp, out = probe
    if int(str(out["versions"][0]).split(".")[0]) >= 2:
        pytest.skip("the second environment's numpy is 2.x too")
    for t, k in enumerate(inp["ap_k"]):
        assert np.array_equal(T.argpartition_c(inp["ap_v%d" % t], int(k) - 1)[:k], out["ap%d" % t]), t


def test_legacy_random_state_stream_is_the_same(probe):
    inp, out = probe
    np.random.seed(4869)
    perm = np.arange(2000)
    for _ in range(3):
        np.random.shuffle(perm)
    assert np.array_equal(perm, out["shuffle3"]) and int(np.random.get_state()[2]) == int(out["shuffle_pos"])


def test_scale_ransac_of_the_installed_sklearn_equals_the_older_release(probe):
    """E_tracker.py:618-636 under the second environment's scikit-learn (0.24.2 here: `base_estimator=`, r2_score nan below two
    samples) against the same call under the test interpreter's: trial count, inlier mask and the position of the global RandomState
    after the fit identical, the coefficient to 1e-12 (two LAPACK builds) -- the RANSAC loop the device restates did not
    change between the releases."""
    from sklearn import linear_model
    import warnings
    inp, out = probe
    assert np.isnan(out["r2_one_sample"]).all() or tuple(out["r2_one_sample"]) == (1.0, 0.0)
    nan_rule = bool(np.isnan(out["r2_one_sample"]).all())
    from oracle import sklearn_compat
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with sklearn_compat.r2_score_like("1.0" if nan_rule else "0.20.3"):
            for i in range(int(inp["rs_n"])):
                ratio = inp["rs_ratio%d" % i]
                np.random.seed(int(inp["rs_seed"][i]))
                r = linear_model.RANSACRegressor(estimator=linear_model.LinearRegression(fit_intercept=False), min_samples=3,
                                                 max_trials=100, stop_probability=0.99, residual_threshold=float(inp["rs_thre"][i]))
                want = out["rs_res%d" % i]
                try:
                    r.fit(ratio.reshape(-1, 1), np.ones((ratio.shape[0], 1)))
                except ValueError:
                    assert want[1] == -1.0, i
                    assert float(np.random.get_state()[2]) == want[3], i
                    continue
                st = np.random.get_state()
                got = np.array([float(r.estimator_.coef_[0, 0]), float(r.n_trials_), float(r.inlier_mask_.sum()), float(st[2]),
                                float(st[1][:8].astype(np.uint64).sum())])
                # the coefficient comes out of each environment's LAPACK (lstsq): equal to rounding; the loop's integers exactly
                assert abs(got[0] - want[0]) <= 1e-12 * abs(want[0]) and np.array_equal(got[1:], want[1:]), (i, got, want)
                assert np.array_equal(r.inlier_mask_, out["rs_mask%d" % i]), i


def test_lanczos_restatement_equals_the_older_pillow(probe):
    """deep_models.py:195-199: the numpy restatement of Pillow's Resample.c (oracle/pil_resample.py, what the device
    reproduces; pinned to the installed Pillow elsewhere) against the second environment's Pillow, bit for bit"""
    inp, out = probe
    for i in range(int(inp["pil_n"])):
        w, h = (int(v) for v in inp["pil_size"][i])
        assert np.array_equal(resize_lanczos_u8(inp["pil_img%d" % i], w, h), out["pil%d" % i]), i


def test_reference_kp_selection_under_numpy_1x_reproduces_the_fixture(tmp_path):
    """the reference's unmodified libs/matching/kp_selection.py executed by the second environment's numpy 1.x (its native
    scalar np.argpartition -- the reference pin's algorithm, no CPU-feature switch) gives the committed fixture
    tests/golden/local_bestN.npz, which make_golden.py wrote under numpy 2.x with the SIMD selection kernels disabled and
    which the oracle restatement and the device reproduce: keypoint values AND order, local_bestN with both score methods"""
    py, vers = _second_python()
    ref = "/root/reference"
    if py is None or not os.path.isdir(ref):
        pytest.skip("needs the second python environment and /root/reference (build container only)")
    if int(vers[0].split(".")[0]) >= 2:
        pytest.skip("the second environment's numpy is 2.x too")
    gold = os.path.join(HERE, "golden", "local_bestN.npz")
    dst = str(tmp_path / "kp.npz")
    r = subprocess.run([py, "-W", "ignore", os.path.join(HERE, "second_env_ref_kp.py"), ref, gold, dst], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    g, o = np.load(gold), np.load(dst)
    n = 0
    for tag in ("a", "b", "c", "d", "ra", "rb", "rc"):
        assert bool(g[tag + "_good"]) == bool(o[tag + "_good"]), tag
        if bool(g[tag + "_good"]):
            assert np.array_equal(g[tag + "_kp1"], o[tag + "_kp1"]) and np.array_equal(g[tag + "_kp2"], o[tag + "_kp2"]), tag
            n += 1
    assert n >= 5
    gb = np.load(os.path.join(HERE, "golden", "bestN.npz"))  # bestN_flow_kp: one argpartition over the whole map
    for tag in "abcd":
        assert np.array_equal(gb[tag + "_kp1"], o["bestN_" + tag + "_kp1"]) and np.array_equal(gb[tag + "_kp2"], o["bestN_" + tag + "_kp2"]), tag
    gg = np.load(os.path.join(HERE, "golden", "gric.npz"))  # the reference's gric.py under numpy 1.x: residuals and both scores
    # the two GRIC scores bit for bit; the 500 epipolar residuals to a few ulp: they come out of BLAS matmuls, whose summation
    # order follows the kernel OpenBLAS dispatches for the HOST CPU (this container has run on hosts where all 500 were equal
    # and on a Xeon where 158 differ by <= 7e-15 relative -- with both scores still equal)
    assert np.allclose(np.asarray(gg["f_res"]), np.asarray(o["f_res"]), rtol=1e-12, atol=0.0)
    for k in ("f_gric", "h_gric"):
        assert np.array_equal(np.asarray(gg[k]), np.asarray(o[k])), k
    # (the homography residual goes through np.linalg.inv and 3x3 matmuls: the two environments' BLAS / LAPACK builds differ
    # in the last bit on a quarter of the points -- 7e-15 absolute; the GRIC score above comes out identical)
    assert np.abs(gg["h_res"] - o["h_res"]).max() <= 1e-12 * np.abs(gg["h_res"]).max()


def test_reference_e_tracker_under_the_older_libraries_reproduces_the_fixture(tmp_path):
    """the reference's unmodified libs/tracker/E_tracker.py -- compute_pose_2d2d (shuffles, 5 x findEssentialMat over the
    oracle's cv2, GRIC, recoverPose) and scale_recovery (triangulation + `RANSACRegressor(base_estimator=...)`, native to this
    scikit-learn: no adapter) -- executed by the second environment (numpy 1.x RandomState / argpartition, scikit-learn 0.24)
    gives tests/golden/e_tracker.npz, written under numpy 2.x / scikit-learn 1.7: poses, inlier masks and the RandomState
    after each case identical, scale to 1e-12 (lstsq of two LAPACK builds)"""
    py, vers = _second_python()
    ref = "/root/reference"
    if py is None or not os.path.isdir(ref):
        pytest.skip("needs the second python environment and /root/reference (build container only)")
    from oracle import cv2_shim
    cv2_shim.lib()  # the C oracle the shim loads is built by the test interpreter
    dst = str(tmp_path / "trk.npz")
    r = subprocess.run([py, "-W", "ignore", os.path.join(HERE, "second_env_ref_tracker.py"), ref, os.path.dirname(HERE), dst],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    g, o = np.load(os.path.join(HERE, "golden", "e_tracker.npz")), np.load(dst)
    print("second environment: numpy %s scikit-learn %s" % tuple(o["versions"]))
    # Inlier masks and RandomState bit for bit.  Poses and scales to 1e-9: they pass through the second environment's BLAS / LAPACK
    # (np.linalg.inv / svd / lstsq, matmuls), whose kernels are picked per HOST CPU -- on the hosts of rounds 3-5 every pose was
    # bit-equal, on a round-6 Xeon they differ by up to 3e-12 (poses) / 2e-11 (scales) with masks and RandomState still equal
    # (round 6: tests/test_oracle_second_env.py was the one CPU test that depended on the machine under the container).
    TOL = 1e-9
    for tag in "abcd":
        assert np.abs(g[tag + "_pose"] - o[tag + "_pose"]).max() <= TOL, tag
        assert np.array_equal(g[tag + "_inliers"], o[tag + "_inliers"]), tag
        assert np.array_equal(g[tag + "_rng_after"], o[tag + "_rng_after"]), tag
        assert abs(float(g[tag + "_scale"]) - float(o[tag + "_scale"])) <= TOL * abs(float(g[tag + "_scale"])), tag
    gf = np.load(os.path.join(HERE, "golden", "e_tracker_flow.npz"))  # validity.method 'flow' (mean-displacement gate)
    for tag in "abcd":
        assert np.abs(gf[tag + "_pose"] - o["flow_" + tag + "_pose"]).max() <= TOL, tag
        for k in ("_inliers", "_rng_after"):
            assert np.array_equal(gf[tag + k], o["flow_" + tag + k]), (tag, k)
    gv = np.load(os.path.join(HERE, "golden", "e_tracker_variants.npz"))  # homo_ratio validity + abs_diff scale RANSAC
    for tag in "abpd":
        assert np.abs(gv[tag + "_pose"] - o["var_" + tag + "_pose"]).max() <= TOL, tag
        for k in ("_inliers", "_rng_after"):
            assert np.array_equal(gv[tag + k], o["var_" + tag + k]), (tag, k)
        a, b = float(gv[tag + "_scale"]), float(o["var_" + tag + "_scale"])
        assert abs(a - b) <= TOL * max(1.0, abs(a)), (tag, a, b)
    gs = np.load(os.path.join(HERE, "golden", "sampled_kp.npz"))  # KeypointSampler.generate_kp_samples + sampled_kp
    for tag in "abc":
        for k in ("_idx", "_kp1", "_kp2"):
            assert np.array_equal(gs[tag + k], o["smp_" + tag + k]), (tag, k)
    gp = np.load(os.path.join(HERE, "golden", "pnp_tracker.npz"))  # the reference's PnpTracker under the same environment
    for tag in "abcde":
        assert np.array_equal(gp[tag + "_kp1"], o["pnp_" + tag + "_kp1"]), tag
        assert np.array_equal(gp[tag + "_rng_after"], o["pnp_" + tag + "_rng_after"]), tag
        assert np.abs(gp[tag + "_pose"] - o["pnp_" + tag + "_pose"]).max() <= 1e-12, tag  # (np.linalg.inv of the final inversion: two LAPACKs)
