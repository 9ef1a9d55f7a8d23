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


def solvePoly(coeffs, maxIters=300):
    """cv2.solvePoly(coeffs[, roots[, maxIters]]) -> retval, roots [n,1,2] (re, im) for real coefficients c[0] + c[1] x + ...
    (core/src/mathfuncs.cpp; the five-point solver calls it with its degree-10 polynomial, five-point.cpp).  retval (the
    last sweep's maxDiff in OpenCV) is not restated: 0.0."""
    c = np.ascontiguousarray(np.asarray(coeffs, np.float64).reshape(-1))
    n = len(c) - 1
    re, im = np.zeros(max(n, 1)), np.zeros(max(n, 1))
    lib().cv3_solve_poly(c, n, re, im, int(maxIters))  # (_dp is an ndpointer: arrays are passed as they are)
    return 0.0, np.stack([re[:n], im[:n]], 1).reshape(n, 1, 2)


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
    """cv2.resize as the reference calls it: INTER_NEAREST on the depth map (dfvo.py:314-317) and the default
    INTER_LINEAR on the loaded uint8 frame (utils.py:51)"""
    src = np.asarray(src)
    if interpolation == INTER_LINEAR:
        return resize_linear_u8(src, dsize)
    assert interpolation == INTER_NEAREST, "oracle cv2 shim implements INTER_NEAREST and 8-bit INTER_LINEAR"
    # nearest: sx = min(floor(x * src_w / dst_w), src_w - 1)
    w, h = dsize
    sh, sw = src.shape[:2]
    ifx = 1.0 / (float(w) / sw)
    ify = 1.0 / (float(h) / sh)
    xs = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), sh - 1)
    return src[ys][:, xs]


def _linear_axis_u8(n_src, n_dst, scale):
    """OpenCV 3.4.3 imgproc/src/resize.cpp, hal::resize, the coefficient loops of the generic path for CV_8U +
    INTER_LINEAR: f = float((d + 0.5) * scale - 0.5), s = floor(f), f -= s (float); the 11-bit fixed-point pair
    saturate_cast<short>((1 - f, f) * 2048) (cvRound: half to even).  Returns s (unclamped) and the [n_dst, 2] table."""
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    return s, f


def resize_linear_u8(src, dsize):
    """cv2.resize(src, (w, h)) for uint8 [H,W] / [H,W,C], interpolation INTER_LINEAR (restated from OpenCV 3.4.3's
    resize.cpp; the IPP branch is skipped there for 8-bit linear, "does not match OpenCV exactly"):
      * scale = 1 / (dst / src) per axis (double);
      * exactly 2 x 2 decimation switches to INTER_AREA's fast path: (a + b + c + d + 2) >> 2;
      * otherwise columns: s < 0 -> (0, f = 0); s >= W - 1 -> (W - 1, f = 0); rows are clipped to [0, H - 1] with
        their weights unchanged; horizontal pass S[s] * a0 + S[s + 1] * a1 (int, 2048 = 1.0), vertical pass
        (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2."""
    src = np.asarray(src)
    assert src.dtype == np.uint8, "8-bit only (the reference resizes the frame right after cv2.imread)"
    w, h = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[:2]
    img = src.reshape(sh, sw, -1).astype(np.int64)
    scale_x, scale_y = 1.0 / (float(w) / sw), 1.0 / (float(h) / sh)
    eps = np.finfo(np.float64).eps
    if abs(scale_x - 2) < eps and abs(scale_y - 2) < eps:
        out = (img[0:2 * h:2, 0:2 * w:2] + img[0:2 * h:2, 1:2 * w:2] + img[1:2 * h:2, 0:2 * w:2] + img[1:2 * h:2, 1:2 * w:2] + 2) >> 2
        return out.astype(np.uint8).reshape((h, w) + src.shape[2:])
    sx, fx = _linear_axis_u8(sw, w, scale_x)
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    sy, fy = _linear_axis_u8(sh, h, scale_y)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    sx1 = np.minimum(sx + 1, sw - 1)  # a1 == 0 wherever this clamps
    rows = img[:, sx] * a0[None, :, None] + img[:, sx1] * a1[None, :, None]         # [H, w, C]
    out = (((b0[:, None, None] * (rows[y0] >> 4)) >> 16) + ((b1[:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8).reshape((h, w) + src.shape[2:])
