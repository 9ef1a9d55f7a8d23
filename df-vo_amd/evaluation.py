"""Trajectory I/O and the KITTI odometry metrics, batched.

What the reference does one pose / one segment at a time in Python -- the trajectory writer of
/root/reference/libs/general/utils.py:329-355 and the evaluator of /root/reference/tools/evaluation/odometry/
kitti_odometry.py:120-299,445-497 (trajectory_distances, calc_sequence_errors over 8 segment lengths from every 10th
frame, compute_overall_err, compute_ATE, compute_RPE) -- is done here on whole arrays: one cumulative sum for the path
length, one searchsorted for all (first frame, length) segments, stacked 4x4 inverses / products for every pose error.
Same definitions, same numbers (tests/test_evaluation_cpu.py: <= 1e-9 against the restatement in oracle/kitti_eval.py, which
is pinned to the reference's own KittiEvalOdom methods and, for the alignment modes, to KittiEvalOdom.eval itself)."""
import numpy as np

LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)


def save_traj(path, poses, fmt="kitti"):
    """one line per frame: "<idx> r11 r12 r13 tx r21 ... tz" (utils.py:329-355, format 'kitti'); str(float) tokens, so the
    file parses back to exactly the same doubles (kitti_odometry.py:95-118)"""
    if fmt != "kitti":
        raise NotImplementedError("save_traj: only the 'kitti' format of the reference's writer is on the hot path")
    poses = np.asarray(poses, np.float64)
    rows = poses[:, :3, :4].reshape(len(poses), 12)
    with open(path, "w") as f:
        f.write("".join("%d %s\n" % (i, " ".join(map(repr, r))) for i, r in enumerate(rows.tolist())))


def load_traj(path):
    """kitti_odometry.py:95-118 load_poses_from_txt for the 13-token lines written above -> [n,4,4]"""
    a = np.loadtxt(path, dtype=np.float64, ndmin=2)
    a = a[:, 1:] if a.shape[1] == 13 else a
    out = np.tile(np.eye(4), (len(a), 1, 1))
    out[:, :3, :4] = a.reshape(len(a), 3, 4)
    return out


def trajectory_distances(poses):
    """cumulative path length at every frame (kitti_odometry.py:120-139); np.cumsum adds in frame order like the loop"""
    p = np.asarray(poses, np.float64)[:, :3, 3]
    step = np.sqrt(((p[:-1] - p[1:]) ** 2).sum(1))
    return np.concatenate([[0.0], np.cumsum(step)])


def _rot_err(pe):
    d = 0.5 * (pe[:, 0, 0] + pe[:, 1, 1] + pe[:, 2, 2] - 1.0)
    return np.arccos(np.clip(d, -1.0, 1.0))


def _trans_err(pe):
    return np.sqrt(pe[:, 0, 3] ** 2 + pe[:, 1, 3] ** 2 + pe[:, 2, 3] ** 2)


def calc_sequence_errors(gt, res, lengths=LENGTHS, step_size=10):
    """kitti_odometry.py:190-245 for all segments at once -> array [k, 5]: first frame, r_err / len, t_err / len, len, speed"""
    gt, res = np.asarray(gt, np.float64), np.asarray(res, np.float64)
    dist = trajectory_distances(gt)
    firsts = np.arange(0, len(gt), step_size)
    F, L = np.meshgrid(firsts, np.asarray(lengths, np.float64), indexing="ij")
    F, L = F.reshape(-1), L.reshape(-1)
    last = np.searchsorted(dist, dist[F] + L, side="right")  # first frame whose distance EXCEEDS dist[first] + len
    ok = (last < len(dist)) & (last < len(res))
    F, L, last = F[ok], L[ok], last[ok]
    d_gt = np.linalg.inv(gt[F]) @ gt[last]
    d_res = np.linalg.inv(res[F]) @ res[last]
    pe = np.linalg.inv(d_res) @ d_gt
    nf = last - F + 1.0
    return np.stack([F.astype(np.float64), _rot_err(pe) / L, _trans_err(pe) / L, L, L / (0.1 * nf)], 1)


ALIGNMENTS = (None, "scale", "scale_7dof", "7dof", "6dof")


def scale_lse_solver(X, Y):
    """kitti_odometry.py:19-31: the s minimising ||s X - Y||"""
    return float(np.sum(X * Y) / np.sum(X ** 2))


def umeyama_alignment(x, y, with_scale=False):
    """kitti_odometry.py:34-84 on stacked arrays: x, y [3, n] -> r [3,3], t [3], c.  The covariance is one matrix product
    instead of n outer products (same sum, different association: agrees to rounding)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    m, n = x.shape
    mean_x, mean_y = x.mean(axis=1), y.mean(axis=1)
    xc, yc = x - mean_x[:, None], y - mean_y[:, None]
    sigma_x = 1.0 / n * (np.linalg.norm(xc) ** 2)
    cov_xy = (yc @ xc.T) * (1.0 / n)
    u, d, v = np.linalg.svd(cov_xy)
    s = np.eye(m)
    if np.linalg.det(u) * np.linalg.det(v) < 0.0:
        s[m - 1, m - 1] = -1
    r = u @ s @ v
    c = float(1 / sigma_x * np.trace(np.diag(d) @ s)) if with_scale else 1.0
    t = mean_y - c * (r @ mean_x)
    return r, t, c


def align(gt, res, alignment=None):
    """What KittiEvalOdom.eval does to the two trajectories before measuring (kitti_odometry.py:618-652): both are moved
    into the frame of their first pose (only the frames the result holds, as the reference's loop does), then
    alignment = None | "scale" (least-squares scale of the positions) | "scale_7dof" (Umeyama's scale only) | "7dof" |
    "6dof" (Umeyama's similarity / rigid transform applied to every pose; "6dof" is the protocol of the README's table).
    Returns (gt, res) as new [n,4,4] arrays."""
    if alignment not in ALIGNMENTS:
        raise ValueError("alignment: None, 'scale', 'scale_7dof', '7dof' or '6dof'")
    gt, res = np.array(gt, np.float64), np.array(res, np.float64)
    n = len(res)
    if len(gt) < n:  # (the reference indexes gt by the result's frame ids and raises KeyError on a missing frame)
        raise ValueError("align: the ground truth holds %d poses, the result %d" % (len(gt), n))
    res = np.linalg.inv(res[0]) @ res
    gt[:n] = np.linalg.inv(gt[0]) @ gt[:n]
    if alignment == "scale":
        res[:, :3, 3] *= scale_lse_solver(res[:, :3, 3], gt[:n, :3, 3])
    elif alignment is not None:
        r, t, c = umeyama_alignment(res[:, :3, 3].T, gt[:n, :3, 3].T, alignment != "6dof")
        res[:, :3, 3] *= c
        if alignment != "scale_7dof":
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = r, t
            res = T @ res
    return gt, res


def evaluate(gt, res, alignment=None, first_frame=True):
    """dict(t_rel [%], r_rel [deg / 100 m], ate [m], rpe_t [m], rpe_r [deg]) of an estimated trajectory against the ground
    truth, both [n,4,4] camera-to-world: KittiEvalOdom.eval's numbers for one sequence (kitti_odometry.py:274-299,445-497,
    618-683) under the chosen alignment.  first_frame=False skips eval()'s first-frame normalisation and the alignment
    (the bare per-method definitions; t_rel / r_rel / RPE do not depend on it, ATE does)."""
    if first_frame:
        gt, res = align(gt, res, alignment)
    elif alignment is not None:
        raise ValueError("alignment needs first_frame=True (the reference aligns after the first-frame normalisation)")
    gt, res = np.asarray(gt, np.float64), np.asarray(res, np.float64)
    err = calc_sequence_errors(gt, res)
    t_rel = float(err[:, 2].mean() * 100.0) if len(err) else 0.0
    r_rel = float(err[:, 1].mean() / np.pi * 180.0 * 100.0) if len(err) else 0.0
    n = len(res)
    ate = float(np.sqrt(np.mean(np.sum((gt[:n, :3, 3] - res[:, :3, 3]) ** 2, 1))))
    g = np.linalg.inv(gt[:n - 1]) @ gt[1:n]
    p = np.linalg.inv(res[:-1]) @ res[1:]
    e = np.linalg.inv(g) @ p
    return {"t_rel": t_rel, "r_rel": r_rel, "ate": ate, "rpe_t": float(_trans_err(e).mean()),
            "rpe_r": float(_rot_err(e).mean() * 180 / np.pi), "segments": int(len(err))}
