"""CPU: pin of the OpenCV-3.4.3 restatement (oracle/cv3_*.c via oracle/cv2_shim.py) to the real library.
tests/golden/opencv343_cases.npz is written by tools/opencv343_dump.py on any machine with opencv-python==3.4.3.18 (there
is none in the build container or on the GPU box, and no network).  While the file is absent this test SKIPS with the
reason "PARITY UNPINNED" -- every "bit-exact vs OpenCV" statement in this repository then means "bit-exact vs the
OpenCV-3.4.3-following oracle"  -- and the independent-property tests in
tests/test_oracle_opencv_properties.py are what stands behind the restatement."""
import os

import numpy as np
import pytest

from oracle import cv2_shim as cv2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv343_cases.npz")


def test_oracle_equals_opencv_343_on_the_dumped_cases():
    if not os.path.exists(GOLD):
        pytest.skip("PARITY UNPINNED: tests/golden/opencv343_cases.npz not present -- run tools/opencv343_dump.py on a machine "
                    "with opencv-python==3.4.3.18 and commit the file; until then the OpenCV subset is checked only through "
                    "independent properties (tests/test_oracle_opencv_properties.py)")
    fx = np.load(GOLD)
    assert str(fx["cv_version"]).startswith("3.4.3"), "dump was made with OpenCV %s" % fx["cv_version"]
    for ci in range(int(fx["n_cases"])):
        p = "c%d_" % ci
        x1, x2, X, K = fx[p + "x1"], fx[p + "x2"], fx[p + "X"], fx[p + "K"]
        f, pp = K[0, 0], (K[0, 2], K[1, 2])
        E, mask = cv2.findEssentialMat(x2, x1, focal=f, pp=pp, method=cv2.RANSAC, prob=0.99, threshold=0.2)
        assert np.array_equal(mask, fx[p + "E_mask"]), "case %d: findEssentialMat inlier mask" % ci
        assert np.abs(E - fx[p + "E"]).max() <= 1e-12
        if (p + "rp_R") in fx.files:
            cnt, R, t, m2 = cv2.recoverPose(fx[p + "E"], x2, x1, focal=f, pp=pp)
            assert cnt == int(fx[p + "rp_cnt"]) and np.array_equal(np.asarray(m2) != 0, fx[p + "rp_mask"] != 0)
            assert np.abs(R - fx[p + "rp_R"]).max() <= 1e-12 and np.abs(t - fx[p + "rp_t"]).max() <= 1e-12
            Kinv = np.linalg.inv(K)
            n = len(x1)
            n1 = (np.c_[x1, np.ones(n)] @ Kinv.T)[:, :2].T.copy()
            n2 = (np.c_[x2, np.ones(n)] @ Kinv.T)[:, :2].T.copy()
            tri = cv2.triangulatePoints(np.c_[np.eye(3), np.zeros(3)], np.c_[fx[p + "rp_R"], fx[p + "rp_t"]], n1, n2)
            assert np.abs(tri / np.linalg.norm(tri, axis=0) - fx[p + "tri"] / np.linalg.norm(fx[p + "tri"], axis=0)).max() <= 1e-9
        H, hm = cv2.findHomography(x2, x1, method=cv2.RANSAC, confidence=0.99, ransacReprojThreshold=1)
        assert np.array_equal(hm, fx[p + "H_mask"]), "case %d: findHomography inlier mask" % ci
        assert np.abs(H - fx[p + "H"]).max() <= 1e-9 * np.abs(fx[p + "H"]).max()
        ok, rvec, tvec, inl = cv2.solvePnPRansac(objectPoints=X, imagePoints=x2, cameraMatrix=K, distCoeffs=None,
                                                 iterationsCount=100, reprojectionError=1)
        assert bool(ok) == bool(fx[p + "pnp_ok"])
        assert np.array_equal(np.asarray(inl).ravel(), fx[p + "pnp_inliers"].ravel()), "case %d: solvePnPRansac inliers" % ci
        assert np.abs(rvec.ravel() - fx[p + "pnp_rvec"].ravel()).max() <= 1e-9
        assert np.abs(tvec.ravel() - fx[p + "pnp_tvec"].ravel()).max() <= 1e-9
        assert np.abs(cv2.Rodrigues(fx[p + "pnp_rvec"])[0] - fx[p + "rod"]).max() <= 1e-12
    for pi in range(int(fx["n_planar"]) if "n_planar" in fx.files else 0):  # coplanar object points: the homography initialisation
        p = "p%d_" % pi
        X, x2, K = fx[p + "X"], fx[p + "x2"], fx[p + "K"]
        ok, rvec, tvec, inl = cv2.solvePnPRansac(objectPoints=X, imagePoints=x2, cameraMatrix=K, distCoeffs=None,
                                                 iterationsCount=100, reprojectionError=1)
        assert bool(ok) == bool(fx[p + "pnp_ok"])
        assert np.array_equal(np.asarray(inl).ravel(), fx[p + "pnp_inliers"].ravel()), "planar case %d: solvePnPRansac inliers" % pi
        assert np.abs(rvec.ravel() - fx[p + "pnp_rvec"].ravel()).max() <= 1e-9
        assert np.abs(tvec.ravel() - fx[p + "pnp_tvec"].ravel()).max() <= 1e-9
    if "five_idx" in fx.files:  # count == modelPoints: the five-point kernel's FIRST model, on 40 five-tuples of case 0
        x1, x2, K = fx["c0_x1"], fx["c0_x2"], fx["c0_K"]
        for ti, idx in enumerate(fx["five_idx"]):
            E5, _ = cv2.findEssentialMat(x2[idx], x1[idx], focal=K[0, 0], pp=(K[0, 2], K[1, 2]), method=cv2.RANSAC, prob=0.99, threshold=0.2)
            want = fx["five_E_%d" % ti]
            assert (E5 is None or E5.size == 0) == (want.size == 0), "five-tuple %d: model count" % ti
            if want.size:
                assert np.abs(np.asarray(E5)[:3] - want[:3]).max() <= 1e-9, "five-tuple %d: first essential matrix" % ti
    for qi in range(int(fx["n_poly"]) if "n_poly" in fx.files else 0):  # cv::solvePoly itself, incl. the coincident-root cases
        _, roots = cv2.solvePoly(fx["poly_c_%d" % qi], maxIters=300)
        got, want = np.asarray(roots).reshape(-1, 2), fx["poly_r_%d" % qi]
        assert got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64)), "solvePoly case %d" % qi
    for ri in range(int(fx["n_resize"]) if "n_resize" in fx.files else 0):
        seed, h, w, oh, ow = [int(v) for v in fx["r%d_spec" % ri]]
        img = fx["r%d_img" % ri]
        assert np.array_equal(cv2.resize(img, (ow, oh)), fx["r%d_linear" % ri]), "resize case %d: 8-bit INTER_LINEAR" % ri
        assert np.array_equal(cv2.resize(img[..., 0].astype(np.float32), (ow, oh), interpolation=cv2.INTER_NEAREST),
                              fx["r%d_nearest" % ri]), "resize case %d: INTER_NEAREST" % ri
