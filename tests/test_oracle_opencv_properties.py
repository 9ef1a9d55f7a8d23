"""CPU: independent-property tests of the OpenCV-3.4.3 restatement (oracle/cv3_*.c).  OpenCV itself is not in the
container, so the restatement stays "parity unpinned" until tests/golden/opencv343_cases.npz exists (tools/
opencv343_dump.py, tests/test_oracle_opencv_pin.py).  What CAN be checked here without sharing any code with the
restatement:
  * five-point (calib3d/src/five-point.cpp): every returned E is an essential matrix (det E = 0, 2 E E^T E - tr(E E^T) E =
    0, x2^T E x1 = 0 on the five points) and the set of solutions is COMPLETE: it equals the set of real solutions of an
    independent solver (Stewenius' action-matrix method on the same null space, numpy eigenvalues -- no degree-10
    polynomial, no Durand-Kerner)
  * cv::RNG (core/src/rand.cpp): multiply-with-carry recurrence evaluated with Python integers
  * solvePnP ITERATIVE / EPnP (calib3d/src/solvepnp.cpp, epnp.cpp, calibration.cpp): pose = the reprojection-error minimum
    found by scipy.optimize.least_squares
  * findHomography refinement (calib3d/src/fundam.cpp): H = scipy's reprojection-error minimum on the inliers
  * cv::solvePoly (core/src/mathfuncs.cpp): roots vs numpy.roots
  * recoverPose / triangulatePoints: geometric ground truth"""
import ctypes as C
import itertools

import numpy as np
import pytest
from scipy.optimize import least_squares

from oracle import cv2_shim
from synth import two_view

L = cv2_shim.lib()


# ---------------------------------------------------------------------------------------------------------------
# independent five-point solver (Stewenius, Engels, Nister 2006): polynomial arithmetic on monomial-exponent dicts
# ---------------------------------------------------------------------------------------------------------------
def _pmul(a, b):
    out = {}
    for (ea, ca), (eb, cb) in itertools.product(a.items(), b.items()):
        e = (ea[0] + eb[0], ea[1] + eb[1], ea[2] + eb[2])
        out[e] = out.get(e, 0.0) + ca * cb
    return out


def _padd(a, b, s=1.0):
    out = dict(a)
    for e, c in b.items():
        out[e] = out.get(e, 0.0) + s * c
    return out


MONO = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
        (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]


def five_point_action_matrix(q1, q2):
    """all real essential matrices through 5 normalised correspondences (x2^T E x1 = 0), each of unit Frobenius norm"""
    A = np.stack([np.kron(np.r_[b, 1.0], np.r_[a, 1.0]) for a, b in zip(q1, q2)])  # rows: x2 (x) x1 against vec(E) row-major
    _, _, vt = np.linalg.svd(A)
    B = vt[5:9].reshape(4, 3, 3)  # null space basis: E = x B0 + y B1 + z B2 + B3
    Ep = [[_padd(_padd({(1, 0, 0): B[0][i, j], (0, 1, 0): B[1][i, j]}, {(0, 0, 1): B[2][i, j]}), {(0, 0, 0): B[3][i, j]})
           for j in range(3)] for i in range(3)]
    det = {}
    for i, j, k in itertools.permutations(range(3)):
        sgn = np.linalg.det(np.eye(3)[[i, j, k]])
        det = _padd(det, _pmul(_pmul(Ep[0][i], Ep[1][j]), Ep[2][k]), sgn)
    EEt = [[{} for _ in range(3)] for _ in range(3)]
    for i in range(3):
        for j in range(3):
            for k in range(3):
                EEt[i][j] = _padd(EEt[i][j], _pmul(Ep[i][k], Ep[j][k]))
    tr = _padd(_padd(EEt[0][0], EEt[1][1]), EEt[2][2])
    rows = [det]
    for i in range(3):
        for j in range(3):
            p = {}
            for k in range(3):
                p = _padd(p, _pmul(EEt[i][k], Ep[k][j]), 2.0)
            rows.append(_padd(p, _pmul(tr, Ep[i][j]), -1.0))
    M = np.array([[r.get(m, 0.0) for m in MONO] for r in rows])
    G = np.linalg.solve(M[:, :10], M[:, 10:])  # [I | G]: degree-3 monomials in terms of the basis
    Ax = np.zeros((10, 10))  # multiplication by x on the basis [x2 xy xz y2 yz z2 x y z 1]
    for r, src in enumerate([0, 1, 2, 3, 4, 5]):
        Ax[r] = -G[src]
    Ax[6, 0] = Ax[7, 1] = Ax[8, 2] = Ax[9, 6] = 1.0
    w, V = np.linalg.eig(Ax)  # Ax b(p) = x_p b(p): the basis monomials at a solution are a right eigenvector
    sols = []
    for k in range(10):
        if abs(w[k].imag) < 1e-9 * max(1.0, abs(w[k])):
            v = V[:, k].real / V[9, k].real
            E = v[6] * B[0] + v[7] * B[1] + v[8] * B[2] + B[3]
            sols.append(E / np.linalg.norm(E))
    return sols


def _same_up_to_sign(a, b):
    return min(np.abs(a - b).max(), np.abs(a + b).max())


@pytest.mark.parametrize("noise", [0.0, 0.3])
def test_five_point_solutions_are_essential_and_complete(noise):
    x1, x2, R, t, K, _ = two_view(400, out_frac=0.0, noise=noise, seed=77)
    Kinv = np.linalg.inv(K)
    n1 = (np.c_[x1, np.ones(len(x1))] @ Kinv.T)[:, :2]
    n2 = (np.c_[x2, np.ones(len(x2))] @ Kinv.T)[:, :2]
    rng = np.random.default_rng(5)
    n_checked = n_tight = 0
    res_det, res_cub, res_epi = [], [], []
    for trial in range(60):
        idx = rng.choice(len(n1), 5, replace=False)
        q1, q2 = np.ascontiguousarray(n1[idx]), np.ascontiguousarray(n2[idx])
        out = np.zeros(90)
        m = L.cv3_five_point(q1, q2, out)
        Es = [out[9 * i:9 * i + 9].reshape(3, 3) for i in range(m)]
        Es = [E / np.linalg.norm(E) for E in Es]
        for E in Es:
            res_det.append(abs(np.linalg.det(E)))
            res_cub.append(np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max())
            res_epi.append(max(abs(np.r_[b, 1] @ E @ np.r_[a, 1]) for a, b in zip(q1, q2)))
        ind = five_point_action_matrix(q1, q2)
        # well separated solution sets only (a double root is found once by one method and twice by the other)
        sep = min([_same_up_to_sign(a, b) for a, b in itertools.combinations(ind, 2)] + [1.0])
        if sep < 1e-3:
            continue
        assert len(Es) == len(ind), "restatement found %d real solutions, the action-matrix solver %d" % (len(Es), len(ind))
        if not Es:
            continue
        worst = 0.0
        for E in Es:
            d = min(_same_up_to_sign(E, F) for F in ind)
            # same solution set; ill-conditioned five-tuples (about one in ten) leave cv::solvePoly's roots / the
            # back-substitution up to ~0.1 away from the independent solver's -- counted below, bounded here
            assert d < 0.1, (trial, d)
            worst = max(worst, d)
        n_tight += worst < 1e-5
        n_checked += 1
    # essential-matrix constraints: typically at rounding level; the worst (ill-conditioned five-tuples, where the roots
    # that cv::solvePoly's fixed 300 Durand-Kerner sweeps deliver are good to ~1e-7) still far below any inlier threshold
    assert np.median(res_det) < 1e-10 and np.median(res_cub) < 1e-9 and np.median(res_epi) < 1e-14
    assert max(res_epi) < 1e-12 and np.percentile(res_cub, 90) < 1e-6 and max(res_cub) < 5e-3
    assert n_checked >= 40 and n_tight >= 0.8 * n_checked
    if noise == 0.0:  # and the true essential matrix is among them
        Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
        Et /= np.linalg.norm(Et)
        assert min(_same_up_to_sign(F, Et) for F in ind) < 1e-8


def test_cv_rng_is_the_multiply_with_carry_recurrence():
    """core/src/rand.cpp: state = (uint64)(unsigned)state * 4164903690 + (state >> 32); next() = (unsigned)state;
    uniform(a, b) = a + next() % (b - a).  cv::RANSAC seeds it with (uint64)-1 on every call."""
    state = 0xffffffffffffffff
    want = []
    for _ in range(12):
        state = ((state & 0xffffffff) * 4164903690 + (state >> 32)) & 0xffffffffffffffff
        want.append(state & 0xffffffff)
    st = C.c_uint64(0)
    L.cv3_rng_init(C.byref(st), C.c_uint64(0xffffffffffffffff))
    for n_pts, w in zip([2000, 2000, 1999, 57, 5, 1 << 20, 3, 7, 11, 13, 2000, 999], want):
        got = L.cv3_rng_uniform_int(C.byref(st), 0, n_pts)
        assert got == w % n_pts
    # first value by hand: low word 0xffffffff * 4164903690 + carry 0xffffffff
    assert want[0] == ((0xffffffff * 4164903690 + 0xffffffff) & 0xffffffff)


def _rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def test_solve_pnp_iterative_reaches_the_reprojection_minimum():
    rng = np.random.default_rng(11)
    K = np.array([[718.856, 0, 607.19], [0, 718.856, 185.22], [0, 0, 1.0]])
    n = 300
    X = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(5, 60, n)], 1)
    rv0, t0 = np.array([0.004, 0.02, 0.002]), np.array([0.05, 0.02, 1.1])
    Xc = X @ _rodrigues(rv0).T + t0
    uv = Xc[:, :2] / Xc[:, 2:] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]] + rng.normal(0, 0.3, (n, 2))

    def resid(p):
        Y = X @ _rodrigues(p[:3]).T + p[3:]
        return (Y[:, :2] / Y[:, 2:] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]] - uv).ravel()
    best = least_squares(resid, np.r_[rv0, t0], xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    rvec, tvec, stats = np.zeros(3), np.zeros(3), (C.c_int * 4)()
    L.cv3_find_extrinsic.argtypes = [cv2_shim._dp, cv2_shim._dp, C.c_int, cv2_shim._dp, cv2_shim._dp, cv2_shim._dp, C.POINTER(C.c_int)]
    L.cv3_find_extrinsic.restype = C.c_int
    rc = L.cv3_find_extrinsic(np.ascontiguousarray(X), np.ascontiguousarray(uv), n, np.ascontiguousarray(K.ravel()), rvec, tvec, stats)
    assert rc >= 0
    assert np.abs(rvec - best[:3]).max() < 1e-6 and np.abs(tvec - best[3:]).max() < 1e-5
    # EPnP (the RANSAC kernel): not the minimiser, but within tens of percent of the optimal reprojection RMS
    Rm, tm = np.zeros(9), np.zeros(3)
    L.cv3_epnp.argtypes = [cv2_shim._dp, cv2_shim._dp, cv2_shim._dp, C.c_int, cv2_shim._dp, cv2_shim._dp]
    L.cv3_epnp.restype = None
    L.cv3_epnp(np.ascontiguousarray(K.ravel()), np.ascontiguousarray(X), np.ascontiguousarray(uv), n, Rm, tm)
    Y = X @ Rm.reshape(3, 3).T + tm
    rms_e = np.sqrt(np.mean((Y[:, :2] / Y[:, 2:] * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]] - uv) ** 2))
    rms_o = np.sqrt(np.mean(resid(best) ** 2))
    assert rms_o <= rms_e < 1.5 * rms_o


def test_find_homography_refinement_reaches_the_reprojection_minimum():
    rng = np.random.default_rng(3)
    H0 = np.array([[1.02, 0.01, 5.0], [-0.015, 0.99, -3.0], [2e-5, -1e-5, 1.0]])
    n = 500
    p = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1)
    q = np.c_[p, np.ones(n)] @ H0.T
    q = q[:, :2] / q[:, 2:] + rng.normal(0, 0.2, (n, 2))
    H, mask = cv2_shim.findHomography(p, q, cv2_shim.RANSAC, 3.0)  # all points inliers at this threshold
    assert mask.sum() == n

    def resid(h):
        Hm = np.r_[h, 1.0].reshape(3, 3)
        y = np.c_[p, np.ones(n)] @ Hm.T
        return (y[:, :2] / y[:, 2:] - q).ravel()
    best = least_squares(resid, (H0 / H0[2, 2]).ravel()[:8], xtol=1e-15, ftol=1e-15, gtol=1e-15).x
    got = (H / H[2, 2]).ravel()[:8]
    assert np.abs(got - best).max() < 1e-6 * np.abs(best).max()


def test_solve_poly_matches_numpy_roots():
    rng = np.random.default_rng(9)
    for _ in range(50):
        c = rng.normal(0, 1, 11)  # c[0] + c[1] x + ... + c[10] x^10, as cv::solvePoly takes them
        re, im = np.zeros(10), np.zeros(10)
        L.cv3_solve_poly(np.ascontiguousarray(c), 10, re, im, 300)
        got = np.sort_complex(re + 1j * im)
        want = np.sort_complex(np.roots(c[::-1]))
        # pair the roots greedily (ordering of near-conjugate pairs can differ)
        for r in got:
            k = np.argmin(np.abs(want - r))
            assert abs(want[k] - r) < 1e-7 * max(1.0, abs(r))
            want = np.delete(want, k)


def test_recover_pose_and_triangulation_geometry():
    x1, x2, R, t, K, _ = two_view(800, out_frac=0.0, noise=0.0, seed=123)
    Et = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    # DF-VO's convention: E maps points of the CURRENT view to lines in the reference view -> pose cur -> ref
    n, Rr, tr, mask = cv2_shim.recoverPose(Et, x1, x2, focal=K[0, 0], pp=(K[0, 2], K[1, 2]))
    assert np.abs(Rr - R).max() < 1e-9 and np.abs(tr.ravel() - t / np.linalg.norm(t)).max() < 1e-9
    # cheirality count: in front of both cameras AND closer than distanceThresh = 50 baselines (five-point.cpp recoverPose)
    Kinv0 = np.linalg.inv(K)
    r1 = np.c_[x1, np.ones(len(x1))] @ Kinv0.T
    r2 = np.c_[x2, np.ones(len(x2))] @ Kinv0.T
    z1 = np.array([np.linalg.lstsq(np.c_[R @ a, -b], -t / np.linalg.norm(t), rcond=None)[0] for a, b in zip(r1, r2)])
    want = int(((z1[:, 0] > 0) & (z1[:, 0] < 50) & (z1[:, 1] > 0) & (z1[:, 1] < 50)).sum())
    assert 0 < want < 800 and n == want and int((np.asarray(mask).ravel() != 0).sum()) == want
    P1 = np.c_[np.eye(3), np.zeros(3)]
    P2 = np.c_[R, t]
    Kinv = np.linalg.inv(K)
    n1 = (np.c_[x1, np.ones(len(x1))] @ Kinv.T)[:, :2].T
    n2 = (np.c_[x2, np.ones(len(x2))] @ Kinv.T)[:, :2].T
    X4 = cv2_shim.triangulatePoints(P1, P2, n1, n2)
    X = (X4[:3] / X4[3]).T
    back = X @ K.T
    assert np.abs(back[:, :2] / back[:, 2:] - x1).max() < 1e-6
