"""CPU: the product's batched trajectory writer / KITTI metrics (df-vo_amd/evaluation.py) against the
restatement of the reference's evaluator in oracle/kitti_eval.py (pinned to the reference's own KittiEvalOdom by
tests/test_oracle_eval.py) on the committed 130-frame trajectories, and the writer against the reference's line format."""
import importlib
import os

import numpy as np

from oracle import kitti_eval as E

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ev():
    import __graft_entry__ as g
    g.dfvo_amd()
    return importlib.import_module("df-vo_amd.evaluation")


def test_batched_metrics_equal_the_reference_definitions():
    ev = _ev()
    fx = np.load(os.path.join(GOLD, "tunnel_traj.npz"))
    gt, res = fx["gt"], fx["poses"]
    want = E.evaluate(list(gt), list(res))
    got = ev.evaluate(gt, res)
    for k in ("t_rel", "r_rel", "ate", "rpe_t", "rpe_r"):
        assert abs(got[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), (k, got[k], want[k])
    e_ref = np.asarray(E.calc_sequence_errors(list(gt), list(res)))
    e_got = ev.calc_sequence_errors(gt, res)
    assert e_got.shape == e_ref.shape and len(e_ref) >= 3
    assert np.array_equal(e_got[:, 0], e_ref[:, 0]) and np.array_equal(e_got[:, 3], e_ref[:, 3])
    assert np.abs(e_got - e_ref).max() <= 1e-12
    assert np.array_equal(ev.trajectory_distances(gt), np.asarray(E.trajectory_distances(list(gt))))
    # a perturbed trajectory (the metrics must move, and still agree)
    rng = np.random.Generator(np.random.PCG64(5))
    res2 = res.copy()
    res2[:, :3, 3] += np.cumsum(rng.normal(0, 0.01, (len(res), 3)), 0)
    w2, g2 = E.evaluate(list(gt), list(res2)), ev.evaluate(gt, res2)
    assert abs(g2["t_rel"] - got["t_rel"]) > 1e-4 and abs(g2["ate"] - got["ate"]) > 1e-4
    for k in ("t_rel", "r_rel", "ate", "rpe_t", "rpe_r"):
        assert abs(g2[k] - w2[k]) <= 1e-9 * max(1.0, abs(w2[k]))


def test_writer_line_format_and_round_trip(tmp_path):
    ev = _ev()
    fx = np.load(os.path.join(GOLD, "tunnel_traj.npz"))
    poses = fx["poses"]
    path = str(tmp_path / "09.txt")
    ev.save_traj(path, poses)
    lines = open(path).read().splitlines()
    assert len(lines) == len(poses)
    for i in (0, 1, len(poses) - 1):  # the reference's writer: str(i) + " " + " ".join(str(j) for j in pose.flatten()[:12])
        assert lines[i] == str(i) + " " + " ".join([str(j) for j in poses[i].flatten()[:12]])
    assert np.array_equal(ev.load_traj(path), poses)
