"""CPU: oracle/sklearn_compat.py -- the one place where the scale-recovery RANSAC of scikit-learn 0.20.3 (the reference's
pin) and of the installed scikit-learn can differ, shown on a crafted input: a one-sample consensus set whose residual is
exactly zero scores 1.0 under the 0.20.3 formula (a later one-sample set with a non-zero residual, score 0.0, then loses
the tie-break) and nan under the >= 0.22 rule (nothing ever loses against nan)."""
import warnings

import numpy as np
import sklearn
from sklearn import linear_model

from oracle.sklearn_compat import _r2_score_020, r2_score_like


def test_r2_of_one_sample():
    assert _r2_score_020([1.0], [1.0]) == 1.0 and _r2_score_020([1.0], [1.5]) == 0.0
    assert _r2_score_020([1.0, 2.0, 4.0], [1.0, 2.0, 4.0]) == 1.0
    want = 1 - ((0.5 ** 2) / (((np.array([1.0, 2.0, 4.0]) - 7 / 3) ** 2).sum()))
    assert _r2_score_020([1.0, 2.0, 4.0], [1.0, 2.0, 4.5]) == want
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with r2_score_like(sklearn.__version__):
            assert np.isnan(sklearn.metrics.r2_score([1.0], [1.0]))
        with r2_score_like("0.20.3"):
            assert sklearn.metrics.r2_score([1.0], [1.0]) == 1.0
        assert np.isnan(sklearn.metrics.r2_score([1.0], [1.0]))  # restored


def test_ransac_tie_break_differs_only_on_an_exact_one_sample_fit():
    # y = 1 (the depth-ratio regression); x: zeros (never inliers: prediction 0), isolated powers of two and two small values.
    # A sample (2^k, 0, 0) fits 1 / 2^k: one inlier, residual exactly 0 -> score 1.0 under the 0.20.3 formula; a sample
    # like (4, 0.5, 0) fits 0.277: one inlier (4 -> 1.108, within 0.3), non-zero residual -> score 0.0, which loses the
    # tie-break against an earlier 1.0.  Under the >= 0.22 rule both score nan and the later set always wins.
    x = np.array([0.0] * 6 + [4.0, 64.0, 1024.0, 16384.0] + [0.5, 0.25]).reshape(-1, 1)
    y = np.ones((x.shape[0], 1))
    coefs = {}
    for version in ("0.20.3", sklearn.__version__):
        got = []
        for seed in range(30):
            np.random.seed(seed)
            r = linear_model.RANSACRegressor(estimator=linear_model.LinearRegression(fit_intercept=False), min_samples=3,
                                             max_trials=40, stop_probability=0.99, residual_threshold=0.3)
            with r2_score_like(version), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    r.fit(x, y)
                    got.append((float(r.estimator_.coef_[0, 0]), int(r.inlier_mask_.sum())))
                except ValueError:
                    got.append(None)
        coefs[version] = got
    major, minor = (int(v) for v in sklearn.__version__.split(".")[:2])
    if major > 0 or minor >= 22:  # the installed rule is the nan rule: some seeds must end on different consensus sets
        assert coefs["0.20.3"] != coefs[sklearn.__version__]
