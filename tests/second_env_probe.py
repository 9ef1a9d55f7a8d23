"""Runs under a SECOND python environment (older numpy / scikit-learn / Pillow than the test interpreter's), with no import
of this repository: evaluates the third-party calls the reference makes on its hot path -- np.argpartition
(kp_selection.py:152-157), sklearn's RANSACRegressor exactly as E_tracker.py:618-636 constructs it (`base_estimator=` where
the library still has that name), numpy's legacy RandomState stream, PIL's LANCZOS resize (deep_models.py:195-199) -- on the
seeded inputs of an .npz and writes what they return.  tests/test_oracle_second_env.py compares the oracle with it.

    <other python> tests/second_env_probe.py in.npz out.npz
"""
import sys
import warnings

import numpy as np


def main(src, dst):
    z = np.load(src)
    out = {}
    import sklearn
    import PIL
    from PIL import Image
    from sklearn import linear_model
    from sklearn.metrics import r2_score
    out["versions"] = np.array([np.__version__, sklearn.__version__, PIL.__version__])
    # np.argpartition(score[mask], k - 1)[:k]
    ks = z["ap_k"]
    for i in range(len(ks)):
        out["ap%d" % i] = np.argpartition(z["ap_v%d" % i], ks[i] - 1)[:ks[i]]
    # legacy RandomState: shuffle (pnp_tracker.py:96, E_tracker.py:218) after the reference's seed
    np.random.seed(4869)
    perm = np.arange(2000)
    for _ in range(3):
        np.random.shuffle(perm)
    out["shuffle3"] = perm
    out["shuffle_pos"] = np.array(np.random.get_state()[2])
    # scale-recovery RANSAC, E_tracker.py:618-636
    kw = {}
    try:
        linear_model.RANSACRegressor(estimator=None)
        kw_name = "estimator"
    except TypeError:
        kw_name = "base_estimator"
    out["ransac_kw"] = np.array(kw_name)
    n = int(z["rs_n"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(n):
            ratio = z["rs_ratio%d" % i]
            np.random.seed(int(z["rs_seed"][i]))
            kw = {kw_name: linear_model.LinearRegression(fit_intercept=False)}
            r = linear_model.RANSACRegressor(min_samples=3, max_trials=100, stop_probability=0.99,
                                             residual_threshold=float(z["rs_thre"][i]), **kw)
            try:
                r.fit(ratio.reshape(-1, 1), np.ones((ratio.shape[0], 1)))
                st = np.random.get_state()
                out["rs_res%d" % i] = np.array([float(r.estimator_.coef_[0, 0]), float(r.n_trials_), float(r.inlier_mask_.sum()),
                                                float(st[2]), float(st[1][:8].astype(np.uint64).sum())])
                out["rs_mask%d" % i] = r.inlier_mask_
            except ValueError as e:  # "RANSAC could not find a valid consensus set"
                out["rs_res%d" % i] = np.array([np.nan, -1.0, -1.0, float(np.random.get_state()[2]), 0.0])
                out["rs_mask%d" % i] = np.zeros(0, bool)
        out["r2_one_sample"] = np.array([r2_score([1.0], [1.0]), r2_score([1.0], [0.5])])
    # PIL LANCZOS
    m = int(z["pil_n"])
    for i in range(m):
        img = z["pil_img%d" % i]
        w, h = int(z["pil_size"][i][0]), int(z["pil_size"][i][1])
        out["pil%d" % i] = np.asarray(Image.fromarray(img).resize((w, h), Image.LANCZOS))
    np.savez(dst, **out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
