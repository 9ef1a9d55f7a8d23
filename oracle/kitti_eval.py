"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the trajectory metrics of /root/reference/tools/evaluation/odometry/kitti_odometry.py:
    :120-139  trajectory_distances          :141-170  rotation_error / translation_error
    :172-188  last_frame_from_segment_length :190-245  calc_sequence_errors (lengths 100 .. 800 m, step 10 frames)
    :274-299  compute_overall_err  (t_rel [%] = ave_t_err * 100, r_rel [deg/100m] = ave_r_err / pi * 180 * 100, :627-628)
    :445-470  compute_ATE           :472-497  compute_RPE
    :19-31    scale_lse_solver      :34-84    umeyama_alignment    :494-517  scale_optimization
    :618-652  eval(): poses <- inv(pose_0) @ pose for both trajectories, then alignment None / "scale" / "scale_7dof" /
              "7dof" / "6dof" (the mode of the README's published table, README.md:101-106)
Pinning: tests/golden/make_golden.py runs the reference's own KittiEvalOdom methods on the committed trajectories;
tests/test_oracle_eval.py checks this module against that fixture (tests/golden/kitti_eval.npz)."""
import numpy as np

LENGTHS = [100, 200, 300, 400, 500, 600, 700, 800]


def trajectory_distances(poses):
    d = [0.0]
    for i in range(len(poses) - 1):
        dx, dy, dz = poses[i][0, 3] - poses[i + 1][0, 3], poses[i][1, 3] - poses[i + 1][1, 3], poses[i][2, 3] - poses[i + 1][2, 3]
        d.append(d[i] + np.sqrt(dx ** 2 + dy ** 2 + dz ** 2))
    return d


def rotation_error(pe):
    d = 0.5 * (pe[0, 0] + pe[1, 1] + pe[2, 2] - 1.0)
    return np.arccos(max(min(d, 1.0), -1.0))


def translation_error(pe):
    return np.sqrt(pe[0, 3] ** 2 + pe[1, 3] ** 2 + pe[2, 3] ** 2)


def calc_sequence_errors(gt, res, lengths=LENGTHS, step_size=10):
    err = []
    dist = trajectory_distances(gt)
    for first in range(0, len(gt), step_size):
        for len_ in lengths:
            last = -1
            for i in range(first, len(dist)):
                if dist[i] > dist[first] + len_:
                    last = i
                    break
            if last == -1 or last >= len(res):
                continue
            d_gt = np.linalg.inv(gt[first]) @ gt[last]
            d_res = np.linalg.inv(res[first]) @ res[last]
            pe = np.linalg.inv(d_res) @ d_gt
            nf = last - first + 1.0
            err.append([first, rotation_error(pe) / len_, translation_error(pe) / len_, len_, len_ / (0.1 * nf)])
    return err


def overall(err):
    """(t_rel [%], r_rel [deg/100m])"""
    if not err:
        return 0.0, 0.0
    t = sum(e[2] for e in err) / len(err)
    r = sum(e[1] for e in err) / len(err)
    return t * 100.0, r / np.pi * 180.0 * 100.0


def ate(gt, res):
    e = [np.sqrt(np.sum((gt[i][:3, 3] - res[i][:3, 3]) ** 2)) for i in range(len(res))]
    return np.sqrt(np.mean(np.asarray(e) ** 2))


def rpe(gt, res):
    """(mean translation RPE [m], mean rotation RPE [deg])"""
    tr, ro = [], []
    for i in range(len(res) - 1):
        g = np.linalg.inv(gt[i]) @ gt[i + 1]
        p = np.linalg.inv(res[i]) @ res[i + 1]
        e = np.linalg.inv(g) @ p
        tr.append(translation_error(e))
        ro.append(rotation_error(e))
    return float(np.mean(tr)), float(np.mean(ro) * 180 / np.pi)


def scale_lse_solver(X, Y):
    return np.sum(X * Y) / np.sum(X ** 2)


def umeyama_alignment(x, y, with_scale=False):
    """x, y [3, n] -> r, t, c  (Umeyama 1991, statement order of kitti_odometry.py:34-84)"""
    m, n = x.shape
    mean_x = x.mean(axis=1)
    mean_y = y.mean(axis=1)
    sigma_x = 1.0 / n * (np.linalg.norm(x - mean_x[:, np.newaxis]) ** 2)
    outer_sum = np.zeros((m, m))
    for i in range(n):
        outer_sum += np.outer((y[:, i] - mean_y), (x[:, i] - mean_x))
    cov_xy = np.multiply(1.0 / n, outer_sum)
    u, d, v = np.linalg.svd(cov_xy)
    s = np.eye(m)
    if np.linalg.det(u) * np.linalg.det(v) < 0.0:
        s[m - 1, m - 1] = -1
    r = u.dot(s).dot(v)
    c = 1 / sigma_x * np.trace(np.diag(d).dot(s)) if with_scale else 1.0
    t = mean_y - np.multiply(c, r.dot(mean_x))
    return r, t, c


def align(gt, res, alignment=None):
    """eval()'s preparation of the two pose lists (kitti_odometry.py:618-652): first-frame alignment of the frames the
    result holds, then the chosen optimisation.  Returns new lists (gt keeps its frames beyond len(res) as they were)."""
    gt = [np.array(p, dtype=np.float64) for p in gt]
    res = [np.array(p, dtype=np.float64) for p in res]
    pred_0, gt_0 = res[0].copy(), gt[0].copy()
    for i in range(len(res)):
        res[i] = np.linalg.inv(pred_0) @ res[i]
        gt[i] = np.linalg.inv(gt_0) @ gt[i]
    if alignment == "scale":
        xyz_pred = np.asarray([p[:3, 3] for p in res])
        xyz_ref = np.asarray([gt[i][:3, 3] for i in range(len(res))])
        scale = scale_lse_solver(xyz_pred, xyz_ref)
        for p in res:
            p[:3, 3] *= scale
    elif alignment in ("scale_7dof", "7dof", "6dof"):
        xyz_gt = np.asarray([[gt[i][0, 3], gt[i][1, 3], gt[i][2, 3]] for i in range(len(res))]).transpose(1, 0)
        xyz_res = np.asarray([[p[0, 3], p[1, 3], p[2, 3]] for p in res]).transpose(1, 0)
        r, t, scale = umeyama_alignment(xyz_res, xyz_gt, alignment != "6dof")
        T = np.eye(4)
        T[:3, :3] = r
        T[:3, 3] = t
        for i in range(len(res)):
            res[i][:3, 3] *= scale
            if alignment in ("7dof", "6dof"):
                res[i] = T @ res[i]
    elif alignment is not None:
        raise ValueError("alignment: None, 'scale', 'scale_7dof', '7dof' or '6dof'")
    return gt, res


def evaluate(gt, res, alignment="none"):
    """alignment="none": the raw lists as given (the per-method definitions); None / "scale" / ... : eval()'s protocol"""
    if alignment != "none":
        gt, res = align(gt, res, alignment)
    t_rel, r_rel = overall(calc_sequence_errors(gt, res))
    rt, rr = rpe(gt, res)
    return {"t_rel": t_rel, "r_rel": r_rel, "ate": float(ate(gt, res)), "rpe_t": rt, "rpe_r": rr}
