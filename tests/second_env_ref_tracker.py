"""Runs under the SECOND python environment (numpy 1.x, scikit-learn < 1.2, no torch): the reference's own
libs/tracker/E_tracker.py, imported from /root/reference unmodified, over the oracle's cv2 shim -- compute_pose_2d2d and
scale_recovery on the seeded cases of tests/golden/e_tracker.npz.  Against make_golden.py's run (numpy 2.x, scikit-learn
1.7 behind a `base_estimator=` -> `estimator=` adapter) this one needs NO adapter: `RANSACRegressor(base_estimator=...)` is
still this scikit-learn's own spelling.  torch is absent here and unused on this path: a stub satisfies the imports of
the RigidFlow layer classes.

    <other python> tests/second_env_ref_tracker.py /root/reference <repo root> out.npz
"""
import os
import sys
import types

import numpy as np


def stub_torch():
    t = types.ModuleType("torch")
    nn = types.ModuleType("torch.nn")
    fn = types.ModuleType("torch.nn.functional")

    class Module:
        def __init__(self, *a, **k):
            pass
    nn.Module = Module
    nn.functional = fn
    t.nn = nn
    t.no_grad = lambda *a, **k: (lambda f: f)
    sys.modules.update({"torch": t, "torch.nn": nn, "torch.nn.functional": fn})


def main(ref, root, dst):
    stub_torch()
    if not hasattr(np, "int"):
        np.int = int  # removed in numpy 1.24 (the reference's pin is 1.16.2): ops_3d.py:29
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle", "ref_shims"))
    sys.path.insert(0, os.path.join(root, "tests"))
    sys.path.insert(0, ref)
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    try:
        import matplotlib.image  # noqa: F401  (libs/general/kitti_utils.py:5; unused on this path)
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.__path__ = []
        for sub in ("pyplot", "image"):
            m = types.ModuleType("matplotlib." + sub)
            setattr(mpl, sub, m)
            sys.modules["matplotlib." + sub] = m
        sys.modules["matplotlib"] = mpl
    from easydict import EasyDict
    from libs.tracker.E_tracker import EssTracker
    from libs.general.timer import Timer
    from libs.geometry.camera_modules import Intrinsics
    import sklearn

    from golden.make_golden_cases import PNP_CASES, pnp_case, tracker_case, variant_case

    cfg = EasyDict({
        "kp_selection": {"rigid_flow_kp": {"enable": False}},
        "e_tracker": {"ransac": {"reproj_thre": 0.2, "repeat": 5}, "validity": {"method": "GRIC", "thre": None},
                      "kp_src": "kp_best", "iterative_kp": {"enable": False}},
        "scale_recovery": {"method": "simple", "kp_src": "kp_best", "iterative_kp": {"enable": False, "kp_src": "kp_depth"},
                           "ransac": {"method": "depth_ratio", "min_samples": 3, "max_trials": 100, "stop_prob": 0.99,
                                      "thre": 0.1}},
        "image": {"height": 376, "width": 1241}})
    out = {"versions": np.array([np.__version__, sklearn.__version__])}
    for tag, (seed, n, of, noise) in {"a": (31, 2000, 0.3, 0.15), "b": (32, 2000, 0.6, 0.3), "c": (33, 600, 0.2, 0.1),
                                      "d": (34, 2000, 0.97, 0.2)}.items():
        c = tracker_case(seed, n, of, noise)
        K = c["K"]
        trk = EssTracker(cfg, Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]]), Timer())
        np.random.seed(4869 + seed)
        res = trk.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], True)
        pose = res["pose"]
        out[tag + "_pose"] = pose.pose.copy()
        out[tag + "_inliers"] = res["inliers"].copy()
        scale = -2.0
        if np.linalg.norm(pose.t) != 0:
            scale = trk.scale_recovery({"kp_best": c["kp_cur"], "depth": c["depth_cur"]}, {"kp_best": c["kp_ref"]}, pose, False)["scale"]
        out[tag + "_scale"] = np.array(float(scale))
        st = np.random.get_state()
        out[tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
    # e_tracker.validity.method 'flow' (ablation_model_sel_flow.yml): the cases of tests/golden/e_tracker_flow.npz
    import copy
    cfg_f = copy.deepcopy(cfg)
    cfg_f.e_tracker.validity = EasyDict({"method": "flow", "thre": 5})
    for tag, (seed, n, of, noise, shrink) in {"a": (51, 2000, 0.3, 0.15, 1.0), "b": (52, 1200, 0.6, 0.3, 1.0),
                                              "c": (53, 2000, 0.2, 0.1, 0.02), "d": (54, 2000, 0.97, 0.2, 1.0)}.items():
        c = tracker_case(seed, n, of, noise)
        kp_ref = c["kp_ref"]
        kp_cur = kp_ref + (c["kp_cur"] - kp_ref) * shrink
        K = c["K"]
        trk = EssTracker(cfg_f, Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]]), Timer())
        np.random.seed(4869 + seed)
        res = trk.compute_pose_2d2d(kp_ref, kp_cur, True)
        st = np.random.get_state()
        out["flow_" + tag + "_pose"], out["flow_" + tag + "_inliers"] = res["pose"].pose.copy(), res["inliers"].copy()
        out["flow_" + tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
    # validity.method 'homo_ratio' + scale_recovery.ransac.method 'abs_diff': the cases of tests/golden/e_tracker_variants.npz
    cfg_v = copy.deepcopy(cfg)
    cfg_v.e_tracker.validity = EasyDict({"method": "homo_ratio", "thre": 0.4})
    cfg_v.scale_recovery.ransac.method = "abs_diff"
    for tag in "abpd":
        c = variant_case(tag)
        K = c["K"]
        trk = EssTracker(cfg_v, Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]]), Timer())
        np.random.seed(4869 + c["seed"])
        res = trk.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], True)
        pose = res["pose"]
        scale = -2.0
        if np.linalg.norm(pose.t) != 0:
            scale = trk.scale_recovery({"kp_best": c["kp_cur"], "depth": c["depth_cur"]}, {"kp_best": c["kp_ref"]}, pose, False)["scale"]
        st = np.random.get_state()
        out["var_" + tag + "_pose"], out["var_" + tag + "_inliers"] = pose.pose.copy(), res["inliers"].copy()
        out["var_" + tag + "_scale"] = np.array(float(scale))
        out["var_" + tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
    # KeypointSampler.generate_kp_samples + sampled_kp (ablation_correspondences_uniform.yml): the cases of sampled_kp.npz
    import libs.matching.kp_selection as kps
    import libs.matching.keypoint_sampler as sampler
    for tag, (h, w, seed, crop, nkp) in {"a": (192, 640, 41, [[0, 1], [0, 1]], 2000), "b": (376, 1241, 42, [[0.1, 0.9], [0.05, 0.95]], 2000),
                                          "c": (61, 83, 43, [[0.3, 1], [0, 0.7]], 37)}.items():
        rng = np.random.Generator(np.random.PCG64(int(seed)))  # make_golden.py:kp_case(h, w, seed, 0.5), its flow output
        rng.random((h, w, 1), dtype=np.float32)
        rng.random((h, w, 1))
        flow = (rng.standard_normal((2, h, w)) * 3).astype(np.float32)
        cfg_s = EasyDict({"kp_selection": {"sampled_kp": {"enable": True, "num_kp": nkp}, "local_bestN": {"enable": False}, "bestN": {"enable": False}},
                          "crop": {"flow_crop": crop}, "image": {"height": h, "width": w}})
        ks = sampler.KeypointSampler(cfg_s)
        xv, yv = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        res = kps.sampled_kp(kp1=kp1, kp2=kp2, ref_data={"depth": np.zeros((h, w))}, kp_list=ks.kps["uniform"], cfg=cfg_s, outputs={})
        out["smp_" + tag + "_idx"] = np.asarray(ks.kps["uniform"], np.int64)
        out["smp_" + tag + "_kp1"], out["smp_" + tag + "_kp2"] = res["kp1_list"], res["kp2_list"]
    # PnpTracker.compute_pose_3d2d (pnp_tracker.py:45-125) on the cases of tests/golden/pnp_tracker.npz
    from libs.tracker.pnp_tracker import PnpTracker
    pcfg = EasyDict({"kp_selection": {"rigid_flow_kp": {"enable": False}}, "depth": {"max_depth": 50.0, "min_depth": 0.0},
                     "pnp_tracker": {"ransac": {"iter": 100, "reproj_thre": 1.0, "repeat": 5}}, "image": {"height": 376, "width": 1241}})
    for tag, (seed, n, of, noise, it, cop) in PNP_CASES.items():
        c = pnp_case(seed, n, of, noise, cop)
        K = c["K"]
        np.random.seed(4869 + seed)
        res = PnpTracker(pcfg, Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])).compute_pose_3d2d(c["kp1"], c["kp2"], c["depth_1"], it)
        st = np.random.get_state()
        out["pnp_" + tag + "_pose"], out["pnp_" + tag + "_kp1"] = res["pose"].pose.copy(), res["kp1"]
        out["pnp_" + tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
    np.savez(dst, **out)


if __name__ == "__main__":
    main(*sys.argv[1:4])
