"""GPU parity of the wavefront-parallel pose solvers (dfvo_find_essential_mat / find_homography /
recover_pose / triangulate_points) against the C oracle (oracle/cv3_*.c) on identical seeded
correspondences.  Bar: inlier masks, RANSAC trajectory (iterations, winning hypothesis) and the
models themselves BIT-EXACT (f64 arithmetic in the same order, no FMA contraction on either side)."""
import ctypes as C

import numpy as np
import pytest

from oracle import cv2_shim as cv2o
from synth import two_view

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trk(gpu):
    lib = gpu.lib()
    t = C.c_void_p()
    gpu.check(lib.dfvo_tracker_create(None, C.byref(t)))
    yield t
    lib.dfvo_tracker_destroy(t)


def hip_E(gpu, trk, x1, x2, K, prob=0.99, thr=0.2, iters=1000):
    n = x1.shape[0]
    E = np.zeros(9)
    mask = np.zeros(max(n, 1), np.uint8)
    info = np.zeros(5, np.int32)
    gpu.check(gpu.lib().dfvo_find_essential_mat(trk, gpu.as_ptr(x1), gpu.as_ptr(x2), n, K[0, 0], K[0, 2], K[1, 2], prob,
                                                thr, iters, gpu.as_ptr(E), gpu.as_ptr(mask), gpu.as_ptr(info)))
    return E.reshape(3, 3), mask[:n], info


CASES = [(2000, 0.3, 0.15, 2002), (2000, 0.5, 0.3, 7), (1500, 0.1, 0.05, 8), (333, 0.6, 0.5, 9), (2000, 0.85, 0.2, 10),
         (64, 0.2, 0.1, 11), (10, 0.0, 0.1, 12)]


@pytest.mark.parametrize("n,out_frac,noise,seed", CASES)
def test_find_essential_mat_bit_exact(gpu, trk, n, out_frac, noise, seed):
    x1, x2, R, t, K, o = two_view(n, out_frac, noise, seed)
    rng = np.random.RandomState(4869 + seed)
    for rep in range(3):  # the reference re-shuffles the points between calls (E_tracker.py:225-228)
        perm = np.arange(n)
        rng.shuffle(perm)
        a, b = np.ascontiguousarray(x1[perm]), np.ascontiguousarray(x2[perm])
        st = {}
        Eo, mo = cv2o.findEssentialMat(a, b, focal=K[0, 0], pp=(K[0, 2], K[1, 2]), method=cv2o.RANSAC, prob=0.99,
                                       threshold=0.2, _stats=st)
        Eh, mh, info = hip_E(gpu, trk, a, b, K)
        print("n=%d rep=%d oracle iters=%d best=(%d,%d) inliers=%s | hip info=%s" % (
            n, rep, st["iters"], st["best_iter"], st["best_model"], None if mo is None else int(mo.sum()), info.tolist()))
        if Eo is None:
            assert info[0] == 0
            continue
        assert info[0] == 1
        assert info[1] == st["iters"] and info[2] == st["best_iter"] and info[3] == st["best_model"]
        assert np.array_equal(mh, mo[:, 0]), "inlier masks differ at %d points" % int((mh != mo[:, 0]).sum())
        assert np.array_equal(Eh, Eo), "E differs: max %g" % np.abs(Eh - Eo).max()


def test_find_essential_mat_config5_8192_hyp_20k_points(gpu, trk):
    x1, x2, R, t, K, o = two_view(20000, 0.3, 0.15, 555, w=1920, h=1280)
    st = {}
    Eo, mo = cv2o.findEssentialMat(x1, x2, focal=K[0, 0], pp=(K[0, 2], K[1, 2]), method=cv2o.RANSAC, prob=0.99,
                                   threshold=0.2, maxIters=8192, _stats=st)
    Eh, mh, info = hip_E(gpu, trk, x1, x2, K, iters=8192)
    print("20k points / 8192 budget: oracle", st, "hip", info.tolist())
    assert info[1] == st["iters"] and info[2] == st["best_iter"] and info[3] == st["best_model"]
    assert np.array_equal(mh, mo[:, 0]) and np.array_equal(Eh, Eo)
    # force the whole hypothesis budget to be evaluated: prob -> 1 keeps niters at the cap
    st = {}
    Eo, mo = cv2o.findEssentialMat(x1[:4000], x2[:4000], focal=K[0, 0], pp=(K[0, 2], K[1, 2]), method=cv2o.RANSAC,
                                   prob=1.0 - 1e-300, threshold=0.2, maxIters=8192, _stats=st)
    Eh, mh, info = hip_E(gpu, trk, np.ascontiguousarray(x1[:4000]), np.ascontiguousarray(x2[:4000]), K,
                         prob=1.0 - 1e-300, iters=8192)
    print("full budget: oracle", st, "hip", info.tolist())
    assert st["iters"] == 8192 and info[1] == 8192
    assert info[2] == st["best_iter"] and info[3] == st["best_model"]
    assert np.array_equal(mh, mo[:, 0]) and np.array_equal(Eh, Eo)


def test_find_essential_mat_too_few_points(gpu, trk):
    x1, x2, R, t, K, o = two_view(4, 0.0, 0.1, 3)
    Eh, mh, info = hip_E(gpu, trk, x1, x2, K)
    assert info[0] == 0 and not mh.any()


@pytest.mark.parametrize("n,out_frac,noise,seed", CASES[:6])
def test_find_homography_bit_exact(gpu, trk, n, out_frac, noise, seed):
    x1, x2, R, t, K, o = two_view(n, out_frac, noise, seed)
    if seed % 2 == 0:  # make half of the cases near-planar so that the homography has real support
        x2 = x1 * np.array([1.01, 0.99]) + np.array([3.0, -1.0]) + (x2 - x1) * 0.02
        x2 = np.ascontiguousarray(x2)
    Ho, mo = cv2o.findHomography(x1, x2, method=cv2o.RANSAC, confidence=0.99, ransacReprojThreshold=1)
    H = np.zeros(9)
    mask = np.zeros(n, np.uint8)
    info = np.zeros(5, np.int32)
    gpu.check(gpu.lib().dfvo_find_homography(trk, gpu.as_ptr(x1), gpu.as_ptr(x2), n, 1.0, 2000, 0.99, gpu.as_ptr(H),
                                             gpu.as_ptr(mask), gpu.as_ptr(info)))
    print("H n=%d: oracle inliers=%s hip info=%s" % (n, None if Ho is None else int(mo.sum()), info.tolist()))
    if Ho is None:
        assert info[0] == 0
        return
    assert info[0] == 1
    assert np.array_equal(mask, mo[:, 0])
    d = np.abs(H.reshape(3, 3) - Ho).max()
    print("   max |H_hip - H_oracle| = %g" % d)
    assert np.array_equal(H.reshape(3, 3), Ho)


@pytest.mark.parametrize("n,out_frac,noise,seed", CASES[:5])
def test_recover_pose_and_triangulation_bit_exact(gpu, trk, n, out_frac, noise, seed):
    x1, x2, R, t, K, o = two_view(n, out_frac, noise, seed)
    Eo, mo = cv2o.findEssentialMat(x1, x2, focal=K[0, 0], pp=(K[0, 2], K[1, 2]), method=cv2o.RANSAC, prob=0.99,
                                   threshold=0.2)
    good_o, Ro, to, mko = cv2o.recoverPose(Eo, x1, x2, focal=K[0, 0], pp=(K[0, 2], K[1, 2]))
    Rh = np.zeros(9)
    th = np.zeros(3)
    mk = np.zeros(n, np.uint8)
    good = C.c_int()
    gpu.check(gpu.lib().dfvo_recover_pose(trk, gpu.as_ptr(np.ascontiguousarray(Eo.reshape(9))), gpu.as_ptr(x1),
                                          gpu.as_ptr(x2), n, K[0, 0], K[0, 2], K[1, 2], gpu.as_ptr(Rh), gpu.as_ptr(th),
                                          gpu.as_ptr(mk), C.byref(good)))
    print("recoverPose n=%d good: oracle %d hip %d" % (n, good_o, good.value))
    assert good.value == good_o
    assert np.array_equal(Rh.reshape(3, 3), Ro) and np.array_equal(th, to[:, 0])
    assert np.array_equal(mk, mko[:, 0])
    # triangulation with [I|0], [R|t] on K-normalised points
    Ki = np.linalg.inv(K)
    a = np.ascontiguousarray((Ki @ np.c_[x1, np.ones(n)].T)[:2])
    b = np.ascontiguousarray((Ki @ np.c_[x2, np.ones(n)].T)[:2])
    P1 = np.ascontiguousarray(np.eye(4)[:3])
    P2 = np.ascontiguousarray(np.c_[Ro, to])
    Xo = cv2o.triangulatePoints(P1, P2, a, b)
    Xh = np.zeros((4, n))
    gpu.check(gpu.lib().dfvo_triangulate_points(trk, gpu.as_ptr(P1), gpu.as_ptr(P2), gpu.as_ptr(a), gpu.as_ptr(b), n,
                                                gpu.as_ptr(Xh)))
    assert np.array_equal(Xh, Xo), "triangulation differs: max %g" % np.abs(Xh - Xo).max()
