"""ORACLE (test infrastructure only): a `cv2`-shaped module over oracle/build/libcv3_oracle.so.

Exposes exactly the cv2 names the reference's hot path touches, with cv2's Python calling
conventions (argument names, return tuples, mask shapes/dtypes), so that the reference's own
E_tracker.py / pnp_tracker.py / ops_3d.py run unmodified on top of it when generating fixtures
(tests/golden/make_golden.py installs this module as `sys.modules['cv2']` in the build container).
PARITY UNPINNED vs real OpenCV 3.4.3 (see oracle/cv3_core.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libcv3_oracle.so")

RANSAC = 8
LMEDS = 4
INTER_NEAREST = 0
INTER_LINEAR = 1
SOLVEPNP_ITERATIVE = 0
COLOR_BGR2RGB = 4

_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    l = C.CDLL(_SO)
    l.cv3_find_essential_mat_ex.restype = C.c_int
    l.cv3_find_essential_mat_ex.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                            C.c_double, C.c_int, _dp, _u8p, _ip, _ip, _ip]
    l.cv3_recover_pose.restype = C.c_int
    l.cv3_recover_pose.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp, _u8p]
    l.cv3_find_homography.restype = C.c_int
    l.cv3_find_homography.argtypes = [_dp, _dp, C.c_int, C.c_double, C.c_int, C.c_double, _dp, _u8p]
    l.cv3_triangulate_points.restype = None
    l.cv3_triangulate_points.argtypes = [_dp, _dp, _dp, _dp, C.c_int, _dp]
    l.cv3_five_point.restype = C.c_int
    l.cv3_five_point.argtypes = [_dp, _dp, _dp]
    l.cv3_decompose_essential_mat.restype = None
    l.cv3_decompose_essential_mat.argtypes = [_dp, _dp, _dp, _dp]
    l.cv3_svd_compute.restype = None
    l.cv3_svd_compute.argtypes = [_dp, C.c_int, C.c_int, _dp, C.c_void_p, C.c_void_p, C.c_int]
    l.cv3_jacobi_eigen.restype = None
    l.cv3_jacobi_eigen.argtypes = [_dp, C.c_int, _dp, _dp]
    l.cv3_invert_lu.restype = C.c_int
    l.cv3_invert_lu.argtypes = [_dp, C.c_int, _dp]
    l.cv3_solve_poly.restype = None
    l.cv3_solve_poly.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int]
    l.cv3_ransac_update_num_iters.restype = C.c_int
    l.cv3_ransac_update_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    l.cv3_rng_init.restype = None
    l.cv3_rng_init.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
    l.cv3_rng_uniform_int.restype = C.c_int
    l.cv3_rng_uniform_int.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
    if hasattr(l, "cv3_solve_pnp_ransac"):
        l.cv3_solve_pnp_ransac.restype = C.c_int
        l.cv3_solve_pnp_ransac.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp, _i32p,
                                           _ip]
        l.cv3_rodrigues_v2m.restype = None
        l.cv3_rodrigues_v2m.argtypes = [_dp, _dp]
        l.cv3_rodrigues_m2v.restype = None
        l.cv3_rodrigues_m2v.argtypes = [_dp, _dp]
    _lib = l
    return l


def _pts(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 2))
    return a


def findEssentialMat(points1, points2, focal=1.0, pp=(0., 0.), method=RANSAC, prob=0.999, threshold=1.0,
                     mask=None, maxIters=1000, _stats=None):
    assert method == RANSAC
    p1, p2 = _pts(points1), _pts(points2)
    n = p1.shape[0]
    E = np.zeros(9)
    m = np.zeros(max(n, 1), np.uint8)
    it, bi, bm = C.c_int(), C.c_int(), C.c_int()
    ok = lib().cv3_find_essential_mat_ex(p1, p2, n, float(focal), float(pp[0]), float(pp[1]), float(prob),
                                         float(threshold), int(maxIters), E, m, C.byref(it), C.byref(bi), C.byref(bm))
    if _stats is not None:
        _stats.update(iters=it.value, best_iter=bi.value, best_model=bm.value)
    if not ok:
        return None, None
    return E.reshape(3, 3), m[:n].reshape(n, 1)


def recoverPose(E, points1, points2, focal=1.0, pp=(0., 0.), mask=None):
    p1, p2 = _pts(points1), _pts(points2)
    n = p1.shape[0]
    R = np.zeros(9)
    t = np.zeros(3)
    m = np.zeros(max(n, 1), np.uint8)
    good = lib().cv3_recover_pose(np.ascontiguousarray(np.asarray(E, np.float64).reshape(9)), p1, p2, n, float(focal),
                                  float(pp[0]), float(pp[1]), R, t, m)
    return good, R.reshape(3, 3), t.reshape(3, 1), m[:n].reshape(n, 1)


def findHomography(srcPoints, dstPoints, method=0, ransacReprojThreshold=3.0, mask=None, maxIters=2000,
                   confidence=0.995):
    assert method == RANSAC
    p1, p2 = _pts(srcPoints), _pts(dstPoints)
    n = p1.shape[0]
    H = np.zeros(9)
    m = np.zeros(max(n, 1), np.uint8)
    ok = lib().cv3_find_homography(p1, p2, n, float(ransacReprojThreshold), int(maxIters), float(confidence), H, m)
    if not ok:
        return None, m[:n].reshape(n, 1)
    return H.reshape(3, 3), m[:n].reshape(n, 1)


def triangulatePoints(projMatr1, projMatr2, projPoints1, projPoints2):
    P1 = np.ascontiguousarray(np.asarray(projMatr1, np.float64).reshape(3, 4))
    P2 = np.ascontiguousarray(np.asarray(projMatr2, np.float64).reshape(3, 4))
    x1 = np.ascontiguousarray(np.asarray(projPoints1, np.float64))
    x2 = np.ascontiguousarray(np.asarray(projPoints2, np.float64))
    n = x1.shape[1]
    X = np.zeros((4, n))
    lib().cv3_triangulate_points(P1, P2, x1, x2, n, X)
    return X


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs, iterationsCount=100, reprojectionError=8.0,
                   confidence=0.99, flags=SOLVEPNP_ITERATIVE):
    assert distCoeffs is None
    obj = np.ascontiguousarray(np.asarray(objectPoints, np.float64).reshape(-1, 3))
    img = _pts(imagePoints)
    n = obj.shape[0]
    K = np.ascontiguousarray(np.asarray(cameraMatrix, np.float64).reshape(9))
    r = np.zeros(3)
    t = np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    cnt = C.c_int()
    ok = lib().cv3_solve_pnp_ransac(obj, img, n, K, int(iterationsCount), float(reprojectionError), float(confidence),
                                    r, t, inl, C.byref(cnt))
    if ok < 0:
        raise NotImplementedError("cv3_solve_pnp_ransac: branch %d of the restatement is not implemented "
                                  "(-2 planar initialisation, -3 four-point P3P kernel)" % ok)
    if not ok:
        return False, r.reshape(3, 1), t.reshape(3, 1), None
    return True, r.reshape(3, 1), t.reshape(3, 1), inl[:cnt.value].reshape(-1, 1).copy()


def Rodrigues(src):
    src = np.asarray(src, np.float64)
    if src.size == 3:
        R = np.zeros(9)
        lib().cv3_rodrigues_v2m(np.ascontiguousarray(src.reshape(3)), R)
        return R.reshape(3, 3), None
    r = np.zeros(3)
    lib().cv3_rodrigues_m2v(np.ascontiguousarray(src.reshape(9)), r)
    return r.reshape(3, 1), None


def resize(src, dsize, interpolation=INTER_LINEAR):
    """nearest-neighbour only (dfvo.py:314-317): sx = min(floor(x * src_w / dst_w), src_w - 1)"""
    assert interpolation == INTER_NEAREST, "oracle cv2 shim implements INTER_NEAREST only"
    src = np.asarray(src)
    w, h = dsize
    sh, sw = src.shape[:2]
    ifx = 1.0 / (float(w) / sw)
    ify = 1.0 / (float(h) / sh)
    xs = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), sh - 1)
    return src[ys][:, xs]
