"""TEST INFRASTRUCTURE (oracle side).  The reference pins scikit-learn 0.20.3 (/root/reference/envs/requirement.yml:233);
this container has a newer one.  The two differ in one place that the scale-recovery RANSAC of
/root/reference/libs/tracker/E_tracker.py:618-636 can reach: sklearn.metrics.r2_score of FEWER THAN TWO samples returns
nan from 0.22 on ("R^2 score is not well-defined with less than two samples"), while 0.20.3 has no such rule and scores the
single sample as a constant target -- 1.0 when the residual is zero, 0.0 otherwise.  RANSACRegressor scores every consensus
set with estimator.score -> r2_score, and a nan best score makes every later trial with the same inlier count win.

r2_score_like("0.20.3") swaps sklearn.metrics.r2_score for a restatement of the 0.20.3 formula (single-output case, the only
one the tracker uses) for the duration of a `with` block; RegressorMixin.score imports r2_score at call time, so the
installed RANSACRegressor picks it up.  PARITY UNPINNED: no scikit-learn 0.20.3 is installable here (no network); the
formula below is the published 0.20.x source restated from memory of its structure -- numerator / denominator sums, a score
of 1 - num / den where both are non-zero, 0.0 for a non-zero numerator over a zero denominator, 1.0 otherwise."""
import contextlib

import numpy as np


def _r2_score_020(y_true, y_pred, sample_weight=None, multioutput="uniform_average", **_):
    y_true = np.asarray(y_true, dtype=np.float64).reshape(len(y_true), -1)
    y_pred = np.asarray(y_pred, dtype=np.float64).reshape(len(y_pred), -1)
    assert sample_weight is None and y_true.shape[1] == 1
    numerator = ((y_true - y_pred) ** 2).sum(axis=0, dtype=np.float64)
    denominator = ((y_true - np.average(y_true, axis=0)) ** 2).sum(axis=0, dtype=np.float64)
    nonzero_denominator = denominator != 0
    nonzero_numerator = numerator != 0
    valid_score = nonzero_denominator & nonzero_numerator
    output_scores = np.ones([y_true.shape[1]])
    output_scores[valid_score] = 1 - (numerator[valid_score] / denominator[valid_score])
    output_scores[nonzero_numerator & ~nonzero_denominator] = 0.0
    return float(np.average(output_scores))


@contextlib.contextmanager
def r2_score_like(version):
    """version "0.20.x" / "0.21.x": the pre-0.22 formula; anything newer: the installed function, untouched"""
    import sklearn.metrics as M
    major, minor = (int(v) for v in str(version).split(".")[:2])
    if major > 0 or minor >= 22:
        yield
        return
    orig = M.r2_score
    M.r2_score = _r2_score_020
    try:
        yield
    finally:
        M.r2_score = orig
