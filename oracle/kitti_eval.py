"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the trajectory metrics of /root/reference/tools/evaluation/odometry/kitti_odometry.py:
    :120-139  trajectory_distances          :141-170  rotation_error / translation_error
    :172-188  last_frame_from_segment_length :190-245  calc_sequence_errors (lengths 100 .. 800 m, step 10 frames)
    :274-299  compute_overall_err  (t_rel [%] = ave_t_err * 100, r_rel [deg/100m] = ave_r_err / pi * 180 * 100, :627-628)
    :445-470  compute_ATE           :472-497  compute_RPE
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


def evaluate(gt, res):
    t_rel, r_rel = overall(calc_sequence_errors(gt, res))
    rt, rr = rpe(gt, res)
    return {"t_rel": t_rel, "r_rel": r_rel, "ate": float(ate(gt, res)), "rpe_t": rt, "rpe_r": rr}
