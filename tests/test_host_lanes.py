"""CPU: the per-lane device functions of the HIP solvers (df-vo_amd/csrc/solver_math.h, np_legacy.h,
kp_select.h), compiled for the host by tests/host_harness, bit-for-bit against the C oracle / numpy.
This is the no-GPU check of the arithmetic the kernels run one lane at a time; the kernels themselves are
compared with the oracle in the -m gpu tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import cv2_shim
from oracle import tracker_np as T

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.fixture(scope="module")
def hh():
    d = os.path.join(HERE, "host_harness")
    subprocess.check_call(["make", "-s", "-C", d])
    lib = C.CDLL(os.path.join(d, "build", "libhost_harness.so"))
    lib.hh_five_point.restype = C.c_int
    lib.hh_ransac_update_num_iters.restype = C.c_int
    lib.hh_ransac_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    return lib


@pytest.fixture(scope="module")
def cv3():
    """a private handle on the oracle library with plain-pointer prototypes"""
    cv2_shim.lib()  # builds it when stale
    lib = C.CDLL(cv2_shim._SO)
    lib.cv3_five_point.restype = C.c_int
    lib.cv3_ransac_update_num_iters.restype = C.c_int
    lib.cv3_ransac_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    return lib


def test_five_point_lane_matches_oracle(hh, cv3):
    rng = np.random.default_rng(11)
    checked = 0
    for trial in range(200):
        q1 = rng.normal(0, 0.3, (5, 2))
        q2 = q1 + rng.normal(0, 0.05, (5, 2))
        e_h, e_o = np.zeros(90), np.zeros(90)
        n_h = hh.hh_five_point(_p(q1), _p(q2), _p(e_h))
        n_o = cv3.cv3_five_point(_p(q1), _p(q2), _p(e_o))
        assert n_h == n_o
        assert np.array_equal(e_h[:9 * n_h], e_o[:9 * n_o])  # bit-exact
        checked += n_h
    assert checked > 200


def _degenerate_five_tuples(rng, kind):
    q1 = rng.normal(0, 0.3, (5, 2))
    q2 = q1 + rng.normal(0, 0.05, (5, 2))
    if kind == 0:  # no motion
        q2 = q1.copy()
    elif kind == 1:  # collinear image points
        t = rng.normal(0, 1, (5, 1))
        q1 = np.hstack([t, 2 * t + 0.1])
        q2 = q1 + rng.normal(0, 0.01, (5, 2))
    elif kind == 2:  # a duplicated correspondence
        q1[1], q2[1] = q1[0], q2[0]
    elif kind == 3:  # un-normalised magnitudes
        q1 = q1 * 1e6
        q2 = q1 + rng.normal(0, 1e4, (5, 2))
    elif kind == 4:
        q1 = q1 * 1e-9
        q2 = q1 + rng.normal(0, 1e-10, (5, 2))
    elif kind == 5:  # pure rotation
        a = 0.05
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        y = np.c_[q1, np.ones(5)] @ R.T
        q2 = y[:, :2] / y[:, 2:]
    elif kind == 6:  # planar scene
        X = np.c_[rng.uniform(-1, 1, (5, 2)), np.full(5, 5.0)]
        Y = X + np.array([0.3, 0.0, 0.1])
        q1, q2 = X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:]
    elif kind == 7:  # a non-finite coordinate
        q2[rng.integers(5), rng.integers(2)] = [np.nan, np.inf, -np.inf][rng.integers(3)]
    elif kind == 8:
        q1, q2 = np.zeros((5, 2)), np.zeros((5, 2))
    elif kind == 9:  # quantised coordinates: exact ties inside the elimination
        q1, q2 = np.round(q1 * 4) / 4, np.round(q2 * 4) / 4
    return np.ascontiguousarray(q1), np.ascontiguousarray(q2)


def test_five_point_lane_on_degenerate_inputs(hh, cv3):
    """the solver the RANSAC kernels run per hypothesis on the inputs a random 5-subset of real matches can be: no motion,
    collinear / duplicated / planar points, pure rotation, huge and tiny magnitudes, NaN / inf, all zeros, quantised
    coordinates -- solution count and every bit of every solution (NaN payloads included) equal to the C oracle's"""
    rng = np.random.default_rng(123)
    for kind in range(10):
        for trial in range(250 if kind != 8 else 2):
            q1, q2 = _degenerate_five_tuples(rng, kind)
            e_h, e_o = np.zeros(90), np.zeros(90)
            n_h = hh.hh_five_point(_p(q1), _p(q2), _p(e_h))
            n_o = cv3.cv3_five_point(_p(q1), _p(q2), _p(e_o))
            assert n_h == n_o, (kind, trial)
            assert np.array_equal(e_h[:9 * n_h].view(np.uint64), e_o[:9 * n_o].view(np.uint64)), (kind, trial)


def test_eig_solvers_match_oracle(hh, cv3):
    rng = np.random.default_rng(12)
    for trial in range(50):
        J = rng.normal(size=(40, 8)) * np.logspace(0, 4, 8)
        A = np.ascontiguousarray(J.T @ J)
        b = rng.normal(size=8)
        x_h, x_o = np.zeros(8), np.zeros(8)
        hh.hh_solve_eig8(_p(A), _p(b), _p(x_h))
        cv3.cv3_solve_eig(_p(A), 8, _p(b), _p(x_o))
        assert np.array_equal(x_h, x_o)
        i_h, i_o = np.zeros(64), np.zeros(64)
        hh.hh_invert_eig8(_p(A), _p(i_h))
        cv3.cv3_invert_eig(_p(A), 8, _p(i_o))
        assert np.array_equal(i_h, i_o)


def test_decompose_and_num_iters_match_oracle(hh, cv3):
    rng = np.random.default_rng(13)
    for trial in range(50):
        E = rng.normal(size=9)
        out_h = [np.zeros(9), np.zeros(9), np.zeros(3)]
        out_o = [np.zeros(9), np.zeros(9), np.zeros(3)]
        hh.hh_decompose_essential(_p(E), *[_p(o) for o in out_h])
        cv3.cv3_decompose_essential_mat(_p(E), *[_p(o) for o in out_o])
        for a, b in zip(out_h, out_o):
            assert np.array_equal(a, b)
    for ep in [0.0, 0.05, 0.3, 0.7, 0.95, 1.0]:
        for mp in (4, 5):
            assert hh.hh_ransac_update_num_iters(0.99, ep, mp, 1000) == cv3.cv3_ransac_update_num_iters(0.99, ep, mp, 1000)


def test_mt_shuffle_lane_matches_numpy(hh):
    for n in (2, 3, 11, 257, 2000):
        rs = np.random.RandomState(4869 + n)
        rs.random_sample(7)  # move off the start of a block
        st = rs.get_state()
        state = np.concatenate([st[1], [st[2]]]).astype(np.uint32)
        perm = np.zeros(n, np.int32)
        hh.hh_mt_shuffle(_p(state, C.c_uint32), n, _p(perm, C.c_int))
        want = np.arange(n)
        rs.shuffle(want)
        assert np.array_equal(perm, want)
        st2 = rs.get_state()
        assert np.array_equal(state[:624], st2[1]) and state[624] == st2[2]


def test_argpartition_lane_matches_oracle(hh):
    rng = np.random.default_rng(3)
    for trial in range(60):
        n = int(rng.integers(1, 5000))
        v = rng.random(n).astype(np.float32)
        if trial % 3 == 0:
            v = np.round(v * 50) / 50
        k = min(20, n)
        tos = np.zeros(n, np.int32)
        hh.hh_argpartition(_p(v, C.c_float), n, k - 1, _p(tos, C.c_int))
        assert np.array_equal(tos[:k], T.argpartition_c(v, k - 1)[:k])
        # the keys-carried-along variant used by k_kp_cell must leave the WHOLE permutation identical
        tos2 = np.zeros(n, np.int32)
        hh.hh_argpartition_cp(_p(v, C.c_float), n, k - 1, _p(tos2, C.c_int))
        assert np.array_equal(tos2, tos)
    for n, k in [(1, 1), (2, 1), (2, 2), (5, 5), (6, 3), (64, 64), (300, 299), (4000, 2000)]:
        v = np.round(rng.random(n).astype(np.float32) * 7) / 7  # many ties
        tos, tos2 = np.zeros(n, np.int32), np.zeros(n, np.int32)
        hh.hh_argpartition(_p(v, C.c_float), n, k - 1, _p(tos, C.c_int))
        hh.hh_argpartition_cp(_p(v, C.c_float), n, k - 1, _p(tos2, C.c_int))
        assert np.array_equal(tos2, tos)


def test_pairwise_sum_lane_matches_numpy(hh):
    """np.add.reduce's pairwise summation (the mean displacement of validity.method 'flow'), every size class: below 8,
    up to the 128-element block, and the recursive split, with magnitudes spread so that the order matters"""
    hh.hh_np_pairwise_sum.restype = C.c_double
    rng = np.random.default_rng(12)
    for n in list(range(0, 20)) + [63, 64, 127, 128, 129, 255, 256, 257, 1000, 1999, 2000, 2001, 4097, 20000]:
        a = np.ascontiguousarray(rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, n))
        got = hh.hh_np_pairwise_sum(_p(a), n)
        want = float(np.add.reduce(a)) if n else 0.0
        assert got == want, (n, got, want)


def test_rigid_flow_lane_matches_oracle(hh):
    """the per-pixel RigidFlow arithmetic of k_rigid_flow_diff (float32, fused multiply-add order of the fixture's GEMM)
    against the oracle restatement, which tests/test_oracle_tracker.py pins to the reference's own layers"""
    from golden.make_golden import rigid_case
    c = rigid_case(120, 200, 63)
    K = c["K"]
    mats = np.ascontiguousarray(np.r_[np.linalg.inv(K).ravel(), c["T_ref_to_cur"].ravel(), K.ravel()].astype(np.float32))
    depth = np.ascontiguousarray(c["raw_depth"], np.float32)
    out = np.zeros((2, 120, 200), np.float32)
    hh.hh_rigid_flow(_p(mats, C.c_float), _p(depth, C.c_float), 120, 200, _p(out, C.c_float))
    want = T.rigid_flow(depth, c["T_ref_to_cur"], K)
    assert np.array_equal(out, want)


def test_lane_parallel_durand_kerner_schedule_is_bit_exact(hh, cv3):
    """df-vo_amd/csrc/solver_poly_lanes.h (one root per lane, updated roots handed over the row in Gauss-Seidel
    order) restated in lock step on the host: same bits as the sequential cv::solvePoly restatement of the oracle and
    as the one-lane device function, on five-point polynomials, random polynomials, repeated / clustered roots and
    polynomials that end in NaN."""
    rng = np.random.default_rng(5)
    polys = []
    for trial in range(3000):  # the polynomials findEssentialMat really solves
        q1 = rng.normal(0, 0.3, (5, 2))
        q2 = q1 + rng.normal(0, 0.05, (5, 2))
        c = np.zeros(11)
        if hh.hh_five_point_poly(_p(q1), _p(q2), _p(c)):
            polys.append(c)
    for trial in range(1000):
        polys.append(rng.normal(0, 1, 11) * 10.0 ** rng.integers(-3, 4, 11))
    for trial in range(200):  # repeated and clustered real roots, conjugate pairs
        r = np.concatenate([np.repeat(rng.normal(0, 1, 3), 2), rng.normal(0, 1e-3, 2) + 0.5, rng.normal(0, 1, 2)])
        polys.append(np.poly(r)[::-1].copy())
    polys.append(np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0]))  # x^10: all denominators hit inf / NaN
    polys.append(np.array([1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1.0]))
    n10 = 0
    for c in polys:
        c = np.ascontiguousarray(c, np.float64)
        if not abs(c[10]) > np.finfo(np.float64).eps:
            continue
        n10 += 1
        a_re, a_im, b_re, b_im, o_re, o_im = (np.zeros(10) for _ in range(6))
        hh.hh_solve_poly10_lockstep(_p(c), _p(a_re), _p(a_im))
        hh.hh_solve_poly10(_p(c), _p(b_re), _p(b_im))
        cv3.cv3_solve_poly(_p(c), 10, _p(o_re), _p(o_im), 300)
        for x, y in ((a_re, b_re), (a_im, b_im), (a_re, o_re), (a_im, o_im)):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64))
    assert n10 > 4000


def test_register_resident_jacobi_schedule_is_bit_exact(hh):
    """df-vo_amd/csrc/h_refine_dev.h jacobi_eigen_coop (round 5: pivot tables and eigenvalues in the owner lanes' registers, the
    pivot maximum as four pairwise exchanges over sixteen lanes with the lower scan order winning ties, one element pair of the
    rotation per lane) restated in lock step on the host: same final matrix, eigenvalues and eigenvectors, bit for bit, as the
    sequential sm::jacobi_eigen_ws (itself pinned to the C oracle by test_eig_solvers_match_oracle) -- on the normal
    matrices of homography refits, random symmetric matrices over 12 decades, matrices full of TIES (equal magnitudes: the
    first maximum in scan order must win), diagonal / zero / rank-one matrices and matrices with NaN or inf entries."""
    rng = np.random.default_rng(17)
    hh.hh_jacobi_eigen_lockstep.restype = C.c_int
    mats = []
    for n in (8, 9):
        for trial in range(400):      # J^T J / L^T L of point sets
            m = rng.integers(n, 40)
            J = rng.normal(0, 1, (m, n)) * 10.0 ** rng.integers(-3, 4, n)
            mats.append(J.T @ J)
        for trial in range(300):      # random symmetric, wide range
            B = rng.normal(0, 1, (n, n)) * 10.0 ** rng.integers(-6, 7)
            mats.append(B + B.T)
        for trial in range(200):      # ties everywhere: small integers, +-1 patterns, constant off-diagonals
            B = rng.integers(-2, 3, (n, n)).astype(np.float64)
            mats.append(B + B.T)
        mats.append(np.ones((n, n)))
        mats.append(np.ones((n, n)) - 2 * np.eye(n))
        mats.append(np.diag(rng.normal(0, 1, n)))
        mats.append(np.zeros((n, n)))
        v = rng.normal(0, 1, n)
        mats.append(np.outer(v, v))
        for trial in range(60):       # NaN / inf somewhere (also in the first candidate's position)
            B = rng.normal(0, 1, (n, n))
            B = B + B.T
            i, j = (0, 1) if trial % 3 == 0 else tuple(sorted(rng.choice(n, 2, replace=False)))
            B[i, j] = B[j, i] = np.nan if trial % 2 == 0 else np.inf
            mats.append(B)
    checked = 0
    for M in mats:
        n = M.shape[0]
        A = np.ascontiguousarray(M, np.float64)
        outs = []
        for fn in (hh.hh_jacobi_eigen_lockstep, hh.hh_jacobi_eigen_sequential):
            Ao, W, V = np.zeros((n, n)), np.zeros(n), np.zeros((n, n))
            with np.errstate(all="ignore"):
                fn(n, _p(A), _p(Ao), _p(W), _p(V))
            outs.append((Ao, W, V))
        for x, y in zip(outs[0], outs[1]):
            assert np.array_equal(x.view(np.uint64), y.view(np.uint64)), "lock-step schedule differs from the sequential Jacobi"
        checked += 1
    assert checked > 1900
