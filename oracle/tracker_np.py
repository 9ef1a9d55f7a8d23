"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the host-side tracker stages of DF-VO, running on top of the C oracle's cv2
subset (oracle/cv2_shim.py) and the installed scikit-learn / numpy RandomState.  Follows (paths
relative to /root/reference):
    libs/matching/kp_selection.py:15-30,74-200        convert_idx_to_global_coord, local_bestN
    libs/matching/keypoint_sampler.py:76-143          kp1 = image_grid, kp2 = kp1 + flow
    libs/general/utils.py:89-114,292-306              preprocess_depth, image_grid
    libs/tracker/gric.py:14-132                       residuals + GRIC
    libs/tracker/E_tracker.py:154-307                 EssTracker.compute_pose_2d2d (validity "GRIC")
    libs/tracker/E_tracker.py:571-643                 EssTracker.find_scale_from_depth
    libs/geometry/ops_3d.py:15-94                     convert_sparse3D_to_depth, triangulation, unprojection_kp
    libs/tracker/pnp_tracker.py:45-125                PnpTracker.compute_pose_3d2d
    libs/dfvo.py:109-262                              hybrid tracking / pose accumulation

Pinning: tests/golden/make_golden.py runs the reference's own kp_selection.py / gric.py / E_tracker.py
(verbatim, cv2 -> oracle shim) on seeded inputs; tests/test_oracle_tracker.py checks this module against
those fixtures.  np.argpartition's order depends on numpy's CPU dispatch (SURVEY.md hard part 3): the
canonical order is the scalar introselect one, restated in `argpartition_scalar` below so that the
oracle does not depend on the host CPU.
"""
import math

import numpy as np

from . import cv2_shim as cv2


# ----------------------------------------------------------------------------------------------
# np.argpartition, scalar introselect order (numpy/core/src/npysort/selection.c.src, argsort flavour)
# ----------------------------------------------------------------------------------------------
def _lt(a, b):
    return a < b or (b != b and a == a)


def argpartition_c(v, kth):
    """np.argpartition(v, kth) in numpy's scalar-introselect order (oracle/np_select.c)"""
    import ctypes as C
    v = np.ascontiguousarray(v, np.float32)
    num = v.shape[0]
    ts = np.arange(num, dtype=np.int64)
    if num == 0:
        return ts
    if kth < 0:
        kth += num
    fn = cv2.lib().np_aintroselect_float
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    fn(v.ctypes.data, ts.ctypes.data, num, kth)
    return ts


def argpartition_scalar(v, kth):
    """pure-python restatement of numpy's scalar `aintroselect` (float keys); returns the index array"""
    v = np.asarray(v)
    num = v.shape[0]
    ts = list(range(num))
    if num == 0:
        return np.zeros(0, np.int64)
    if kth < 0:
        kth += num
    _introselect(v, ts, 0, num, kth)
    return np.asarray(ts, np.int64)


def _introselect(v, ts, base, num, kth):
    """operates on ts[base:base+num], kth relative to base"""
    low, high = 0, num - 1

    def V(i):
        return v[ts[base + i]]

    def swap(a, b):
        ts[base + a], ts[base + b] = ts[base + b], ts[base + a]

    if kth - low < 3:
        for i in range(0, kth + 1):
            minidx, minval = i, V(i)
            for k in range(i + 1, num):
                if _lt(V(k), minval):
                    minidx, minval = k, V(k)
            swap(i, minidx)
        return
    if kth == num - 1:
        maxidx, maxval = low, V(low)
        for k in range(low + 1, num):
            if not _lt(V(k), maxval):
                maxidx, maxval = k, V(k)
        swap(kth, maxidx)
        return
    depth_limit = (int(num).bit_length() - 1) * 2
    while low + 1 < high:
        ll, hh = low + 1, high
        if depth_limit > 0 or hh - ll < 5:
            mid = low + (high - low) // 2
            if _lt(V(high), V(mid)):
                swap(high, mid)
            if _lt(V(high), V(low)):
                swap(high, low)
            if _lt(V(low), V(mid)):
                swap(low, mid)
            swap(mid, low + 1)
        else:
            raise NotImplementedError("median-of-medians fallback not needed for the oracle's inputs")
        depth_limit -= 1
        pivot = V(low)
        while True:
            ll += 1
            while _lt(V(ll), pivot):
                ll += 1
            hh -= 1
            while _lt(pivot, V(hh)):
                hh -= 1
            if hh < ll:
                break
            swap(hh, ll)
        swap(low, hh)
        if hh >= kth:
            high = hh - 1
        if hh <= kth:
            low = ll
    if high == low + 1:
        if _lt(V(high), V(low)):
            swap(high, low)


# ----------------------------------------------------------------------------------------------
# keypoint selection
# ----------------------------------------------------------------------------------------------
def image_grid(h, w):
    """utils.py:292-306"""
    xv, yv = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
    return np.transpose(np.stack([xv, yv]), (1, 2, 0))


def local_bestN(flow, flow_diff, num_bestN=2000, num_row=10, num_col=10, thre=0.1, argpartition=argpartition_c,
                score_method="flow"):
    """kp_selection.py:74-200 with score_method 'flow' or 'flow_ratio' (:137-141,151-156), depth consistency disabled.
    flow [2,H,W] f32, flow_diff [H,W,1] f32 -> dict(good_kp_found, kp1_best [1,N,2], kp2_best [1,N,2])"""
    h, w, _ = flow_diff.shape
    kp1 = np.expand_dims(image_grid(h, w), 0)
    kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
    n_best = math.floor(num_bestN / (num_col * num_row))
    diff = np.expand_dims(flow_diff, 0)
    out = {"good_kp_found": True}
    if (diff[0, :, :, 0] < thre).sum() < num_bestN * 0.1:
        out["good_kp_found"] = False
        return out
    sel, good_region_cnt = [], 0
    for row in range(num_row):
        for col in range(num_col):
            x0 = [int(h / num_row * row), int(w / num_col * col)]
            x1 = [int(h / num_row * (row + 1)) - 1, int(w / num_col * (col + 1)) - 1]
            tile = diff[:, x0[0]:x1[0], x0[1]:x1[1]].copy()
            if score_method == "flow_ratio":
                tmp_flow = np.transpose(np.expand_dims(flow[:, x0[0]:x1[0], x0[1]:x1[1]], 0), (0, 2, 3, 1))
                with np.errstate(divide="ignore", invalid="ignore"):
                    tile = tile / np.linalg.norm(tmp_flow, axis=3, keepdims=True)
            where = np.where(tile < thre)
            num_to_pick = min(n_best, len(where[0]))
            if num_to_pick != 0:
                good_region_cnt += 1
            order = argpartition(tile[where], num_to_pick - 1)[:num_to_pick]
            for i in order:
                sel.append((where[1][i] + x0[0], where[2][i] + x0[1]))
    if good_region_cnt < (num_row * num_col) * 0.1:
        out["good_kp_found"] = False
        return out
    ys = np.asarray([s[0] for s in sel], np.int64)
    xs = np.asarray([s[1] for s in sel], np.int64)
    out["kp1_best"] = kp1[:, ys, xs]
    out["kp2_best"] = kp2[:, ys, xs]
    return out


def bestN_flow_kp(flow, flow_diff, N=2000, argpartition=argpartition_c):
    """kp_selection.py:33-71.  flow [2,H,W] f32, flow_diff [H,W,1] f32 -> kp1_best, kp2_best [1,N,2]"""
    h, w, _ = flow_diff.shape
    kp1 = np.expand_dims(image_grid(h, w), 0)
    kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
    diff = np.expand_dims(flow_diff, 0)
    where = np.where(diff >= 0)
    sel = argpartition(diff[where], N)[:N]
    ys, xs = where[1][sel], where[2][sel]
    return kp1[:, ys, xs], kp2[:, ys, xs]


def generate_kp_samples(img_h, img_w, crop, N):
    """keypoint_sampler.py:51-74: N indices spread uniformly over the cropped grid (row-major)"""
    y0, y1 = int(crop[0][0] * img_h), int(crop[0][1] * img_h)
    x0, x1 = int(crop[1][0] * img_w), int(crop[1][1] * img_w)
    total_num = (x1 - x0) * (y1 - y0) - 1
    return np.linspace(0, total_num, N, dtype=int)


def sampled_kp(flow, kp_list, crop=None):
    """kp_selection.py:327-378.  flow [2,H,W] f32 -> kp1_list, kp2_list [1,N,2] at the indices kp_list of the cropped grid"""
    h, w = flow.shape[1:]
    kp1 = np.expand_dims(image_grid(h, w), 0)
    kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
    if crop is not None:
        y0, y1 = int(h * crop[0][0]), int(h * crop[0][1])
        x0, x1 = int(w * crop[1][0]), int(w * crop[1][1])
        kp1, kp2 = kp1[:, y0:y1, x0:x1], kp2[:, y0:y1, x0:x1]
    a = np.transpose(kp1.reshape(1, -1, 2), (1, 0, 2))[kp_list]
    b = np.transpose(kp2.reshape(1, -1, 2), (1, 0, 2))[kp_list]
    return np.transpose(a, (1, 0, 2)), np.transpose(b, (1, 0, 2))


def _dot_fma_f32(coefs, vals):
    """float32 dot product the way the fixtures' torch-CPU GEMM evaluates it: a0*b0 rounded, then fused multiply-adds in
    ascending k.  fma(a, b, c) is emulated through float64 (the product of two float32 is exact there)."""
    acc = None
    for a, v in zip(coefs, vals):
        if acc is None:
            acc = (np.float32(a) * v).astype(np.float32)
        else:
            acc = (np.float64(a) * v.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def rigid_flow(raw_depth, T, K):
    """RigidFlow layer (geometry/rigid_flow.py:17-58 = Backprojection (backprojection.py:56-62) -> Transformation3D
    (transformation3d.py:29) -> Projection(normalized=False) (projection.py:46-52) -> PixToFlow (layers.py:262)) in
    float32, as E_tracker.py:672-683 calls it.  raw_depth [H,W] f32, T [4,4], K [3,3] -> [2,H,W] f32.
    torch.matmul's float32 rounding depends on the BLAS kernel the host CPU selects (the same torch build gives
    different last bits on the build container and on the GPU box), so the three small matmuls are written out with the
    operation order of the machine that produced tests/golden/rigid_flow_kp.npz by running the reference itself;
    tests/test_oracle_tracker.py pins this restatement to that fixture bit for bit."""
    f32 = np.float32
    h, w = raw_depth.shape
    K4, iK4 = np.eye(4), np.eye(4)
    K4[:3, :3] = K
    iK4[:3, :3] = np.linalg.inv(K)
    K4, iK4, Tf = K4.astype(f32), iK4.astype(f32), np.asarray(T).astype(f32)
    xx, yy = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
    one = np.ones_like(xx)
    d = np.ascontiguousarray(raw_depth, dtype=f32)
    P = [(d * _dot_fma_f32(iK4[r, :3], [xx, yy, one])).astype(f32) for r in range(3)] + [one]
    Q = [_dot_fma_f32(Tf[r], P) for r in range(4)]
    U = [_dot_fma_f32(K4[r], Q) for r in range(3)]
    den = (U[2] + f32(1e-7)).astype(f32)
    return np.stack([((U[0] / den).astype(f32) - xx).astype(f32), ((U[1] / den).astype(f32) - yy).astype(f32)])


def opt_rigid_flow_kp(flow, flow_diff, rigid_flow_diff, score_method="opt_flow", num_bestN=2000, num_row=10, num_col=10,
                      rigid_thre=5, opt_thre=0.1, argpartition=argpartition_c):
    """kp_selection.py:203-324.  flow [2,H,W] f32, flow_diff / rigid_flow_diff [H,W,1] f32 ->
    dict(kp1_depth, kp2_depth, kp1_depth_uniform, kp2_depth_uniform [1,N,2], rigid_flow_mask [H,W])"""
    h, w, _ = rigid_flow_diff.shape
    kp1 = np.expand_dims(image_grid(h, w), 0)
    kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
    n_best = math.floor(num_bestN / (num_col * num_row))
    rdiff = np.expand_dims(rigid_flow_diff, 0)
    odiff = np.expand_dims(flow_diff, 0)
    best, uni = [], []
    for row in range(num_row):
        for col in range(num_col):
            x0 = [int(h / num_row * row), int(w / num_col * col)]
            x1 = [int(h / num_row * (row + 1)) - 1, int(w / num_col * (col + 1)) - 1]
            to = odiff[:, x0[0]:x1[0], x0[1]:x1[1]].copy()
            tr = rdiff[:, x0[0]:x1[0], x0[1]:x1[1]].copy()
            mask = (tr < rigid_thre) * (to < opt_thre)
            score = tr if score_method == "rigid_flow" else to
            where = np.where(mask)
            cnt = len(where[0])
            num_to_pick = min(n_best, cnt)
            if num_to_pick > 0:
                step = int(cnt / num_to_pick)
                sel = np.arange(0, cnt, step)[:num_to_pick]
            else:
                sel = []
            for i in sel:
                uni.append((where[1][i] + x0[0], where[2][i] + x0[1]))
            order = argpartition(score[where], num_to_pick - 1)[:num_to_pick]
            for i in order:
                best.append((where[1][i] + x0[0], where[2][i] + x0[1]))
    assert len(best) != 0, "sampling threshold is too small."
    out = {}
    for name, sel in (("depth", best), ("depth_uniform", uni)):
        ys = np.asarray([s[0] for s in sel], np.int64)
        xs = np.asarray([s[1] for s in sel], np.int64)
        out["kp1_" + name] = kp1[:, ys, xs].copy()
        out["kp2_" + name] = kp2[:, ys, xs].copy()
    out["rigid_flow_mask"] = rdiff[0, :, :, 0]
    return out


def kp_selection_good_depth(flow, flow_diff, raw_depth, rigid_flow_pose, K, score_method="opt_flow", **kw):
    """E_tracker.py:645-705: rigid flow of the reference depth under `rigid_flow_pose`, its distance to the optical flow,
    then opt_rigid_flow_kp"""
    rf = rigid_flow(raw_depth, rigid_flow_pose, K)
    rdiff = np.linalg.norm(rf - flow, axis=0)
    return opt_rigid_flow_kp(flow, flow_diff, np.expand_dims(rdiff, 2), score_method, **kw)


def scale_recovery_iterative(flow, flow_diff, raw_depth, depth_cur, E_pose, K, prev_scale=0, score_method="opt_flow",
                             kp_src="kp_depth", kp_best=None, **kw):
    """E_tracker.py:509-569.  E_pose: 4x4 cur -> ref with unit translation.  Returns dict(scale, cur_kp, ref_kp,
    n_iter); consumes np.random through find_scale_from_depth."""
    scale = prev_scale
    out = {}
    for it in range(5):
        P = np.array(E_pose, dtype=np.float64, copy=True)
        P[:3, 3] *= scale
        sel = kp_selection_good_depth(flow, flow_diff, raw_depth, np.linalg.inv(P), K, score_method, **kw)
        ref_kp_depth, cur_kp_depth = sel["kp1_depth_uniform"][0], sel["kp2_depth_uniform"][0]
        if kp_src == "kp_depth":
            ref_kp, cur_kp = ref_kp_depth, cur_kp_depth
        else:
            ref_kp, cur_kp = kp_best
        new_scale = find_scale_from_depth(ref_kp, cur_kp, np.linalg.inv(np.asarray(E_pose, np.float64)), depth_cur, K)
        delta = np.abs(new_scale - scale)
        scale = new_scale
        out = {"scale": scale, "cur_kp": cur_kp_depth, "ref_kp": ref_kp_depth, "rigid_flow_mask": sel["rigid_flow_mask"],
               "n_iter": it + 1}
        if delta < 0.001:
            return out
    return out


def preprocess_depth(depth, crop, depth_range):
    """utils.py:89-114"""
    min_depth, max_depth = depth_range
    h, w = depth.shape
    y0, y1 = int(h * crop[0][0]), int(h * crop[0][1])
    x0, x1 = int(w * crop[1][0]), int(w * crop[1][1])
    m = np.zeros((h, w))
    m[y0:y1, x0:x1] = 1
    return depth * (m * ((depth < max_depth) * (depth > min_depth)))


# ----------------------------------------------------------------------------------------------
# GRIC
# ----------------------------------------------------------------------------------------------
def fundamental_residual(F, kp1, kp2):
    """gric.py:14-37 (per-point form of the same expression)"""
    m0 = np.c_[kp1, np.ones(len(kp1))].T
    m1 = np.c_[kp2, np.ones(len(kp2))].T
    Fm0 = F @ m0
    Ftm1 = F.T @ m1
    m1Fm0 = (Fm0 * m1).sum(0)
    return m1Fm0 ** 2 / (np.sum(Fm0[:2] ** 2, axis=0) + np.sum(Ftm1[:2] ** 2, axis=0))


def homography_residual(H_in, kp1, kp2):
    """gric.py:40-92"""
    H = H_in.flatten()
    m0x, m0y, m1x, m1y = kp1[:, 0], kp1[:, 1], kp2[:, 0], kp2[:, 1]
    G0 = np.stack([H[0] - m1x * H[6], H[1] - m1x * H[7], -m0x * H[6] - m0y * H[7] - H[8]])
    G1 = np.stack([H[3] - m1y * H[6], H[4] - m1y * H[7], -m0x * H[6] - m0y * H[7] - H[8]])
    magG0 = np.sqrt(G0[0] * G0[0] + G0[1] * G0[1] + G0[2] * G0[2])
    magG1 = np.sqrt(G1[0] * G1[0] + G1[1] * G1[1] + G1[2] * G1[2])
    alpha = np.arccos((G0[0] * G1[0] + G0[1] * G1[1]) / (magG0 * magG1))
    alg0 = m0x * H[0] + m0y * H[1] + H[2] - m1x * (m0x * H[6] + m0y * H[7] + H[8])
    alg1 = m0x * H[3] + m0y * H[4] + H[5] - m1y * (m0x * H[6] + m0y * H[7] + H[8])
    D1, D2 = alg0 / magG0, alg1 / magG1
    return (D1 * D1 + D2 * D2 - 2.0 * D1 * D2 * np.cos(alpha)) / np.sin(alpha)


def calc_GRIC(res, sigma, n, model):
    """gric.py:95-132"""
    R = 4
    K = {"FMat": 7, "EMat": 5, "HMat": 8}[model]
    D = {"FMat": 3, "EMat": 3, "HMat": 2}[model]
    lam3RD = 2.0 * (R - D)
    s = 0
    for i in range(n):
        tmp = res[i] * (1. / sigma ** 2)
        s += tmp if tmp <= lam3RD else lam3RD
    return s + n * D * np.log(R) + K * np.log(R * n)


# ----------------------------------------------------------------------------------------------
# EssTracker
# ----------------------------------------------------------------------------------------------
def compute_pose_2d2d(kp_ref, kp_cur, K, reproj_thre=0.2, repeat=5, max_iters=1000, validity="GRIC", validity_thre=None):
    """E_tracker.py:154-307, validity.method 'GRIC' (default configuration) or 'flow' (ablation_model_sel_flow.yml).
    Consumes np.random (global RandomState).  Returns dict(R, t, inliers, and diagnostics)."""
    if validity == "flow":
        return _compute_pose_2d2d_flow(kp_ref, kp_cur, K, reproj_thre, repeat, max_iters, validity_thre)
    if validity == "homo_ratio":
        return _compute_pose_2d2d_homo_ratio(kp_ref, kp_cur, K, reproj_thre, repeat, max_iters, validity_thre)
    assert validity == "GRIC"
    fx, cx, cy = K[0, 0], K[0, 2], K[1, 2]
    n = kp_ref.shape[0]
    R, t = np.eye(3), np.zeros((3, 1))
    best_cnt = 0
    best_inliers = np.ones((n, 1)) == 1
    diag = {"rep_inliers": [], "rep_valid": [], "rep_gric": [], "num_valid": 0, "major_valid": False,
            "cheirality": 0, "h_gric": None}
    valid_case = True
    if n > 10:
        H, H_inl = cv2.findHomography(kp_cur, kp_ref, method=cv2.RANSAC, confidence=0.99, ransacReprojThreshold=1)
        H_gric = calc_GRIC(homography_residual(H, kp_cur, kp_ref), 0.8, n, "HMat")
        diag["h_gric"] = H_gric
    else:
        valid_case = False
    best_E = None
    if valid_case:
        num_valid = 0
        for _ in range(repeat):
            new_list = np.arange(0, n, 1)
            np.random.shuffle(new_list)
            a, b = kp_cur.copy()[new_list], kp_ref.copy()[new_list]
            E, inl = cv2.findEssentialMat(a, b, focal=fx, pp=(cx, cy), method=cv2.RANSAC, prob=0.99,
                                          threshold=reproj_thre, maxIters=max_iters)
            F = np.linalg.inv(K.T) @ E @ np.linalg.inv(K)
            E_gric = calc_GRIC(fundamental_residual(F, a, b), 0.8, n, "EMat")
            valid_case = H_gric > E_gric
            diag["rep_inliers"].append(int(inl.sum()))
            diag["rep_valid"].append(bool(valid_case))
            diag["rep_gric"].append(float(E_gric))
            if inl.sum() > best_cnt:
                best_E, best_cnt = E, inl.sum()
                revert = np.zeros_like(new_list)
                for cnt, i in enumerate(new_list):
                    revert[i] = cnt
                best_inliers = inl[list(revert)]
            num_valid += valid_case * 1
        diag["num_valid"] = int(num_valid)
        if num_valid > (repeat / 2):
            diag["major_valid"] = True
            good, Rr, tr, _ = cv2.recoverPose(best_E, kp_cur, kp_ref, focal=fx, pp=(cx, cy))
            diag["cheirality"] = int(good)
            if good > n * 0.1:
                R, t = Rr, tr
    out = {"R": R, "t": t, "inliers": best_inliers[:, 0] == 1, "best_inlier_cnt": int(best_cnt)}
    out.update(diag)
    return out


def _compute_pose_2d2d_flow(kp_ref, kp_cur, K, reproj_thre, repeat, max_iters, thre):
    """E_tracker.py:182-185 (mean flow magnitude gate), :243-250 (per-repeat recoverPose cheirality as the validity and
    as a second condition of the best-model update), :281-300 (unchanged ending)"""
    fx, cx, cy = K[0, 0], K[0, 2], K[1, 2]
    n = kp_ref.shape[0]
    R, t = np.eye(3), np.zeros((3, 1))
    best_cnt = 0
    best_inliers = np.ones((n, 1)) == 1
    diag = {"rep_inliers": [], "rep_valid": [], "rep_cheirality": [], "num_valid": 0, "major_valid": False,
            "cheirality": 0}
    avg_flow = np.mean(np.linalg.norm(kp_ref - kp_cur, axis=1))
    diag["avg_flow"] = float(avg_flow)
    valid_case = avg_flow > thre
    best_E = None
    if valid_case:
        num_valid = 0
        for _ in range(repeat):
            new_list = np.arange(0, n, 1)
            np.random.shuffle(new_list)
            a, b = kp_cur.copy()[new_list], kp_ref.copy()[new_list]
            E, inl = cv2.findEssentialMat(a, b, focal=fx, pp=(cx, cy), method=cv2.RANSAC, prob=0.99,
                                          threshold=reproj_thre, maxIters=max_iters)
            cheir, _, _, _ = cv2.recoverPose(E, a, b, focal=fx, pp=(cx, cy))
            valid_case = cheir > n * 0.1
            diag["rep_inliers"].append(int(inl.sum()))
            diag["rep_valid"].append(bool(valid_case))
            diag["rep_cheirality"].append(int(cheir))
            if inl.sum() > best_cnt and cheir > n * 0.05:
                best_E, best_cnt = E, inl.sum()
                revert = np.zeros_like(new_list)
                for cnt, i in enumerate(new_list):
                    revert[i] = cnt
                best_inliers = inl[list(revert)]
            num_valid += valid_case * 1
        diag["num_valid"] = int(num_valid)
        if num_valid > (repeat / 2):
            diag["major_valid"] = True
            good, Rr, tr, _ = cv2.recoverPose(best_E, kp_cur, kp_ref, focal=fx, pp=(cx, cy))
            diag["cheirality"] = int(good)
            if good > n * 0.1:
                R, t = Rr, tr
    out = {"R": R, "t": t, "inliers": best_inliers[:, 0] == 1, "best_inlier_cnt": int(best_cnt)}
    out.update(diag)
    return out


def _compute_pose_2d2d_homo_ratio(kp_ref, kp_cur, K, reproj_thre, repeat, max_iters, thre):
    """E_tracker.py:186-194 (findHomography, ransacReprojThreshold 0.2), :243-250 (per repeat: the share of homography
    inliers among homography + essential inliers below `thre`), :276-300 (unchanged ending)"""
    fx, cx, cy = K[0, 0], K[0, 2], K[1, 2]
    n = kp_ref.shape[0]
    R, t = np.eye(3), np.zeros((3, 1))
    best_cnt = 0
    best_inliers = np.ones((n, 1)) == 1
    diag = {"rep_inliers": [], "rep_valid": [], "rep_ratio": [], "num_valid": 0, "major_valid": False, "cheirality": 0}
    H, H_inl = cv2.findHomography(kp_cur, kp_ref, method=cv2.RANSAC, confidence=0.99, ransacReprojThreshold=0.2)
    diag["h_inliers"] = int(H_inl.sum())
    best_E = None
    num_valid = 0
    for _ in range(repeat):
        new_list = np.arange(0, n, 1)
        np.random.shuffle(new_list)
        a, b = kp_cur.copy()[new_list], kp_ref.copy()[new_list]
        E, inl = cv2.findEssentialMat(a, b, focal=fx, pp=(cx, cy), method=cv2.RANSAC, prob=0.99,
                                      threshold=reproj_thre, maxIters=max_iters)
        with np.errstate(invalid="ignore"):
            ratio = H_inl.sum() / (H_inl.sum() + inl.sum())
        valid_case = ratio < thre
        diag["rep_inliers"].append(int(inl.sum()))
        diag["rep_valid"].append(bool(valid_case))
        diag["rep_ratio"].append(float(ratio))
        if inl.sum() > best_cnt:
            best_E, best_cnt = E, inl.sum()
            revert = np.zeros_like(new_list)
            for cnt, i in enumerate(new_list):
                revert[i] = cnt
            best_inliers = inl[list(revert)]
        num_valid += valid_case * 1
    diag["num_valid"] = int(num_valid)
    if num_valid > (repeat / 2):
        diag["major_valid"] = True
        good, Rr, tr, _ = cv2.recoverPose(best_E, kp_cur, kp_ref, focal=fx, pp=(cx, cy))
        diag["cheirality"] = int(good)
        if good > n * 0.1:
            R, t = Rr, tr
    out = {"R": R, "t": t, "inliers": best_inliers[:, 0] == 1, "best_inlier_cnt": int(best_cnt)}
    out.update(diag)
    return out


def convert_sparse3D_to_depth(kp, XYZ, height, width):
    """ops_3d.py:15-41"""
    depth = np.zeros((height, width))
    kp_int = kp.astype(int)
    y_idx = (kp_int[:, 0] >= 0) * (kp_int[:, 0] < width)
    kp_int = kp_int[y_idx]
    x_idx = (kp_int[:, 1] >= 0) * (kp_int[:, 1] < height)
    kp_int = kp_int[x_idx]
    XYZ = XYZ[:, y_idx]
    XYZ = XYZ[:, x_idx]
    depth[kp_int[:, 1], kp_int[:, 0]] = XYZ[2]
    return depth


def triangulation(kp1, kp2, T_1w, T_2w):
    """ops_3d.py:44-67"""
    X = cv2.triangulatePoints(T_1w[:3], T_2w[:3], np.ascontiguousarray(kp1.T), np.ascontiguousarray(kp2.T))
    X /= X[3]
    return X[:3], T_1w[:3] @ X, T_2w[:3] @ X


def find_scale_from_depth(kp1, kp2, T_21, depth2, K, min_samples=3, max_trials=100, stop_prob=0.99, thre=0.1,
                          diag=None, method="depth_ratio"):
    """E_tracker.py:571-643 with ransac.method 'depth_ratio' or 'abs_diff'.  Consumes np.random through sklearn."""
    from sklearn import linear_model
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    img_h, img_w = depth2.shape
    kp1n, kp2n = kp1.copy(), kp2.copy()
    kp1n[:, 0] = (kp1[:, 0] - cx) / fx
    kp1n[:, 1] = (kp1[:, 1] - cy) / fy
    kp2n[:, 0] = (kp2[:, 0] - cx) / fx
    kp2n[:, 1] = (kp2[:, 1] - cy) / fy
    with np.errstate(all="ignore"):
        _, _, X2 = triangulation(kp1n, kp2n, np.eye(4), T_21)
        tri = convert_sparse3D_to_depth(kp2, X2, img_h, img_w)
        tri[tri < 0] = 0
    valid = (depth2 > 0) * (tri > 0)
    ratio = tri[valid] / depth2[valid]
    if diag is not None:
        diag["n_valid"] = int(valid.sum())
        diag["ratios"] = ratio
    if valid.sum() > 10:
        ransac = linear_model.RANSACRegressor(estimator=linear_model.LinearRegression(fit_intercept=False),
                                              min_samples=min_samples, max_trials=max_trials,
                                              stop_probability=stop_prob, residual_threshold=thre)
        if method == "depth_ratio":
            ransac.fit(ratio.reshape(-1, 1), np.ones((ratio.shape[0], 1)))
        else:
            assert method == "abs_diff"
            ransac.fit(tri[valid].reshape(-1, 1), depth2[valid].reshape(-1, 1))
        if diag is not None:
            diag["n_trials"] = int(ransac.n_trials_)
            diag["n_inliers"] = int(ransac.inlier_mask_.sum())
        return float(ransac.estimator_.coef_[0, 0])
    return -1


# ----------------------------------------------------------------------------------------------
# PnpTracker
# ----------------------------------------------------------------------------------------------
def unprojection_kp(kp, kp_depth, inv_K):
    """ops_3d.py:70-94"""
    N = kp.shape[0]
    XYZ = np.ones((N, 3, 1))
    XYZ[:, :2, 0] = kp
    inv_K_b = np.ones((1, 3, 3))
    inv_K_b[0] = inv_K
    inv_K_b = np.repeat(inv_K_b, N, axis=0)
    XYZ = np.matmul(inv_K_b, XYZ)[:, :, 0]
    XYZ[:, 0] = XYZ[:, 0] * kp_depth
    XYZ[:, 1] = XYZ[:, 1] * kp_depth
    XYZ[:, 2] = XYZ[:, 2] * kp_depth
    return XYZ


def compute_pose_3d2d(kp1, kp2, depth_1, K, min_depth=0.0, max_depth=50.0, repeat=5, iters=100, reproj_thre=1.0):
    """pnp_tracker.py:45-125.  Consumes np.random (one shuffle per repeat).  Returns dict(pose 4x4 (view-2 ->
    view-1, i.e. after the final inversion), R, t of the best solvePnPRansac, kp1, kp2, XYZ, best_inlier)."""
    height, width = depth_1.shape
    x_idx = (kp2[:, 0] >= 0) * (kp2[:, 0] < width)
    kp1 = kp1[x_idx]
    kp2 = kp2[x_idx]
    y_idx = (kp2[:, 1] >= 0) * (kp2[:, 1] < height)
    kp1 = kp1[y_idx]
    kp2 = kp2[y_idx]
    kp1_int = kp1.astype(int)
    kp_depths = depth_1[kp1_int[:, 1], kp1_int[:, 0]]
    non_zero_mask = (kp_depths != 0)
    depth_range_mask = (kp_depths < max_depth) * (kp_depths > min_depth)
    valid_kp_mask = non_zero_mask * depth_range_mask
    kp1 = kp1[valid_kp_mask]
    kp2 = kp2[valid_kp_mask]
    XYZ_kp1 = unprojection_kp(kp1, kp_depths[valid_kp_mask], np.linalg.inv(K))
    best_rt = []
    best_inlier = 0
    for _ in range(repeat):
        new_list = np.arange(0, kp2.shape[0], 1)
        np.random.shuffle(new_list)
        new_XYZ = XYZ_kp1.copy()[new_list]
        new_kp2 = kp2.copy()[new_list]
        if new_kp2.shape[0] > 4:
            flag, r, t, inlier = cv2.solvePnPRansac(objectPoints=new_XYZ, imagePoints=new_kp2, cameraMatrix=K,
                                                    distCoeffs=None, iterationsCount=iters,
                                                    reprojectionError=reproj_thre)
            if flag and inlier.shape[0] > best_inlier:
                best_rt = [r, t]
                best_inlier = inlier.shape[0]
    pose = np.eye(4)
    R, t = np.eye(3), np.zeros((3, 1))
    if len(best_rt) != 0:
        r, t = best_rt
        R = cv2.Rodrigues(r)[0]
        pose[:3, :3] = R
        pose[:3, 3:] = np.asarray(t).reshape(3, 1)
    pose = np.linalg.inv(pose)
    return {"pose": pose, "R": R, "t": np.asarray(t).reshape(3, 1), "kp1": kp1, "kp2": kp2, "XYZ": XYZ_kp1,
            "best_inlier": best_inlier}
