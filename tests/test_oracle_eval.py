"""CPU: oracle/kitti_eval.py (restatement of the reference's trajectory metrics) against the fixture produced by the
reference's own KittiEvalOdom methods (tests/golden/make_golden.py -> kitti_eval.npz), and -- when /root/reference is
present (build container) -- against the reference code live."""
import os
import sys
import types

import numpy as np

from oracle import kitti_eval as E

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_eval.npz")


def _check(fx):
    gt, res = list(fx["gt"]), list(fx["res"])
    err = np.array(E.calc_sequence_errors(gt, res))
    assert err.shape == fx["seq_err"].shape and np.allclose(err, fx["seq_err"], rtol=0, atol=1e-15)
    t_rel, r_rel = E.overall(E.calc_sequence_errors(gt, res))
    assert abs(t_rel - float(fx["t_rel"])) <= 1e-12 and abs(r_rel - float(fx["r_rel"])) <= 1e-12
    assert abs(E.ate(gt, res) - float(fx["ate"])) <= 1e-12
    rt, rr = E.rpe(gt, res)
    assert abs(rt - float(fx["rpe_t"])) <= 1e-12 and abs(rr - float(fx["rpe_r"])) <= 1e-12


def test_kitti_eval_matches_reference_fixture():
    _check(np.load(GOLD))


def test_kitti_eval_matches_reference_live():
    ref = "/root/reference/tools/evaluation/odometry/kitti_odometry.py"
    if not os.path.exists(ref):
        import pytest
        pytest.skip("reference not present (GPU box): the committed fixture pins the restatement")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as G
    fx = np.load(GOLD)
    live = G.kitti_eval_reference(list(fx["gt"]), list(fx["res"]))
    for k in ("t_rel", "r_rel", "ate", "rpe_t", "rpe_r"):
        assert abs(live[k] - float(fx[k])) <= 1e-12


ALIGN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_eval_align.npz")
MODES = [None, "scale", "scale_7dof", "7dof", "6dof"]
CASES = ["tunnel", "drive", "drive_short"]


def test_alignment_modes_match_reference_eval_fixture():
    """oracle align() + metrics against KittiEvalOdom.eval itself (kitti_odometry.py:556-700) run by make_golden.py on three
    trajectory pairs (non-identity first poses, a result shorter than the ground truth) under all five alignment modes"""
    fx = np.load(ALIGN)
    for name in CASES:
        gt, res = list(fx[name + "_gt"]), list(fx[name + "_res"])
        for al in MODES:
            got = E.evaluate(gt, res, alignment=al)
            want = fx["%s_%s" % (name, al)]
            for k, w in zip(("t_rel", "r_rel", "ate", "rpe_t", "rpe_r"), want):
                # the trajectories travelled through a text file with repr() precision: exact doubles, so 1e-9 is loose
                assert abs(got[k] - float(w)) <= 1e-9 * max(1.0, abs(float(w))), (name, al, k, got[k], float(w))
    # the modes are not all the same thing
    a = fx["drive_None"]
    assert abs(fx["drive_scale"][0] - a[0]) > 0.1 and abs(fx["drive_7dof"][2] - a[2]) > 1.0 and abs(fx["drive_6dof"][2] - a[2]) > 1.0


def test_alignment_matches_reference_live():
    if not os.path.exists("/root/reference/tools/evaluation/odometry/kitti_odometry.py"):
        import pytest
        pytest.skip("reference not present (GPU box): the committed fixture pins the restatement")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as G
    fx = np.load(ALIGN)
    live = G.kitti_eval_reference_eval(fx["drive_short_gt"], fx["drive_short_res"], "6dof")
    for k, w in zip(("t_rel", "r_rel", "ate", "rpe_t", "rpe_r"), fx["drive_short_6dof"]):
        assert abs(live[k] - float(w)) <= 1e-12
