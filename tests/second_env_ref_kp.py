"""Runs under the SECOND python environment (numpy 1.x): the reference's own libs/matching/kp_selection.py, imported from
/root/reference unmodified, on the seeded cases of tests/golden/local_bestN.npz.  numpy 1.x's np.argpartition is the scalar
introselect the reference's pinned numpy runs -- no NPY_DISABLE_CPU_FEATURES needed here.

    <other python> tests/second_env_ref_kp.py /root/reference local_bestN.npz out.npz
"""
import importlib.util
import os
import sys

import numpy as np


class Cfg(dict):  # attribute access like EasyDict, which kp_selection.py expects of its cfg
    def __getattr__(self, k):
        v = self[k]
        return Cfg(v) if isinstance(v, dict) else v


def kp_case(h, w, seed, frac):  # tests/golden/make_golden.py:kp_case, verbatim (numpy-version independent: PCG64 Generator)
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    diff = rng.random((int(h), int(w), 1), dtype=np.float32) * np.float32(0.1 / frac)
    diff[rng.random((int(h), int(w), 1)) < 0.01] = np.float32(0.05)
    flow = (rng.standard_normal((2, int(h), int(w))) * 3).astype(np.float32)
    return diff, flow


def main(ref, gold, dst):
    spec = importlib.util.spec_from_file_location("ref_kp_selection", os.path.join(ref, "libs/matching/kp_selection.py"))
    kps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kps)
    g = np.load(gold)
    out = {"numpy": np.array(np.__version__)}
    for tag in ("a", "b", "c", "d", "ra", "rb", "rc"):
        sp = g[tag + "_spec"]
        ratio = tag.startswith("r")
        h, w, seed, frac = int(sp[0]), int(sp[1]), int(sp[2]), float(sp[3])
        diff, flow = kp_case(h, w, seed, frac)
        lb = {"enable": True, "num_bestN": 2000, "num_row": 10, "num_col": 10, "score_method": "flow", "thre": 0.1}
        if ratio:
            flow[:, 5:9, 7:30] = 0
            diff[6, 10:14, 0] = 0
            lb["score_method"], lb["thre"] = "flow_ratio", float(sp[4])
        cfg = Cfg({"kp_selection": {"local_bestN": lb, "depth_consistency": {"enable": False, "thre": 0.05}}})
        xv, yv = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            res = kps.local_bestN(kp1=kp1, kp2=kp2, ref_data={"flow_diff": diff, "flow": flow}, cfg=cfg, outputs={"good_kp_found": True})
        out[tag + "_good"] = np.array(res["good_kp_found"])
        if res["good_kp_found"]:
            out[tag + "_kp1"], out[tag + "_kp2"] = res["kp1_best"], res["kp2_best"]
    # bestN_flow_kp (ablation_correspondences_best_n.yml): ONE np.argpartition over the whole image, cases with heavy ties / NaNs
    cases = {"a": (192, 640, 71, 0.6, 2000, 0), "b": (376, 1241, 72, 0.35, 2000, 0), "c": (60, 90, 73, 0.5, 300, 0),
             "d": (120, 200, 74, 0.5, 1000, 1)}  # tests/golden/make_golden.py:BESTN_CASES
    for tag, (h, w, seed, frac, N, hard) in cases.items():
        diff, flow = kp_case(h, w, seed, frac)
        if hard:
            rng = np.random.Generator(np.random.PCG64(int(seed) + 1))
            diff = (np.round(diff * 400) / 400).astype(np.float32)
            diff[rng.random(diff.shape) < 0.002] = np.float32("nan")
        cfg = Cfg({"kp_selection": {"bestN": {"enable": True, "num_bestN": N}}})
        xv, yv = np.meshgrid(np.linspace(0, w - 1, w), np.linspace(0, h - 1, h))
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        with np.errstate(invalid="ignore"):
            res = kps.bestN_flow_kp(kp1=kp1, kp2=kp2, ref_data={"flow_diff": diff}, cfg=cfg, outputs={})
        out["bestN_" + tag + "_kp1"], out["bestN_" + tag + "_kp2"] = res["kp1_best"], res["kp2_best"]
    # libs/tracker/gric.py (pure numpy) on the inputs stored in tests/golden/gric.npz
    gp = os.path.join(os.path.dirname(gold), "gric.npz")
    if os.path.exists(gp):
        spec = importlib.util.spec_from_file_location("ref_gric", os.path.join(ref, "libs/tracker/gric.py"))
        gric = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gric)
        gg = np.load(gp)
        f_res = gric.compute_fundamental_residual(gg["F"], gg["kp1"], gg["kp2"])
        h_res = gric.compute_homography_residual(gg["H"], gg["kp1"], gg["kp2"])
        n = gg["kp1"].shape[0]
        out["f_res"], out["h_res"] = f_res, h_res
        out["f_gric"], out["h_gric"] = np.array(gric.calc_GRIC(f_res, 0.8, n, "EMat")), np.array(gric.calc_GRIC(h_res, 0.8, n, "HMat"))
    np.savez(dst, **out)


if __name__ == "__main__":
    main(*sys.argv[1:4])
