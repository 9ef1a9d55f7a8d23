"""Generate golden fixtures by running the REFERENCE's own Python (read from /root/reference) in
the build container.  The reference cannot travel to the GPU box, so the outputs are committed
under tests/golden/*.npz together with this script.

    python tests/golden/make_golden.py            # (re)writes the fixtures

What runs here is reference code verbatim; only third-party imports are shimmed
(oracle/ref_shims/README.md) and three documented compat patches are applied:
  * correlation.FunctionCorrelation (cupy CUDA string)  -> oracle.nets_torch.correlation
  * F.grid_sample default                              -> align_corners=True (torch 1.1 semantics)
  * Tensor.cuda / Module.cuda / torch.cuda.current_stream -> CPU no-ops; np.int -> int
np.argpartition order is CPU-dispatch dependent (SURVEY.md hard part 3): run with
NPY_DISABLE_CPU_FEATURES set (done below by re-exec) so that the scalar introselect order is the
canonical one.
"""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

_SIMD_OFF = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"
if __name__ == "__main__" and os.environ.get("NPY_DISABLE_CPU_FEATURES") is None and "--no-reexec" not in sys.argv:
    env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=_SIMD_OFF)
    os.execve(sys.executable, [sys.executable] + sys.argv + ["--no-reexec"], env)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shims"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

from oracle import nets_torch as O  # noqa: E402


def apply_compat():
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=0)
    torch.cuda.manual_seed = lambda *a, **k: None

    def _is_cuda(d):
        return (isinstance(d, str) and d.startswith("cuda")) or (isinstance(d, torch.device) and d.type == "cuda")
    _mto, _tto = torch.nn.Module.to, torch.Tensor.to

    def module_to(self, *a, **k):  # .to(torch.device('cuda')) -> stay on the CPU
        a = tuple(x for x in a if not _is_cuda(x))
        k = {kk: v for kk, v in k.items() if not (kk == "device" and _is_cuda(v))}
        return _mto(self, *a, **k) if (a or k) else self

    def tensor_to(self, *a, **k):
        a = tuple(x for x in a if not _is_cuda(x))
        k = {kk: v for kk, v in k.items() if not (kk == "device" and _is_cuda(v))}
        return _tto(self, *a, **k) if (a or k) else self
    torch.nn.Module.to = module_to
    torch.Tensor.to = tensor_to
    _load = torch.load

    def load_cpu(f, map_location=None, **k):  # torch.load(path, map_location=torch.device('cuda')) -> CPU; old pickles
        k.setdefault("weights_only", False)
        return _load(f, map_location="cpu", **k)
    torch.load = load_cpu
    _gs = F.grid_sample

    def grid_sample(input, grid, mode='bilinear', padding_mode='zeros', align_corners=None):
        return _gs(input, grid, mode=mode, padding_mode=padding_mode,
                   align_corners=True if align_corners is None else align_corners)
    F.grid_sample = grid_sample
    torch.nn.functional.grid_sample = grid_sample


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def golden_liteflownet():
    from libs.deep_models.flow.lite_flow_net import correlation as ref_corr
    ref_corr.FunctionCorrelation = lambda tensorFirst, tensorSecond, intStride: O.correlation(
        tensorFirst, tensorSecond, intStride)
    from libs.deep_models.flow.lite_flow_net import lite_flow_net as ref_lfn
    net = ref_lfn.LiteFlowNet()
    sd = O.liteflownet_state_dict(seed=4869)
    missing = net.load_state_dict(sd, strict=True)
    print("LiteFlowNet load_state_dict:", missing)
    net.eval()
    g = torch.Generator().manual_seed(1001)
    h, w = 64, 96
    base = torch.rand(2, 3, h + 8, w + 8, generator=g)
    base = F.avg_pool2d(base, 5, 1, 2)  # smooth texture
    first = base[:, :, 4:4 + h, 4:4 + w].contiguous()
    second = base[:, :, 3:3 + h, 6:6 + w].contiguous()  # shifted copy
    with torch.no_grad():
        flows = net([first, second])
    out = {"first": first.numpy(), "second": second.numpy()}
    for k, v in flows.items():
        out["flow%d" % k] = v.numpy()
        print("  flow", k, tuple(v.shape), float(v.abs().max()))
    np.savez_compressed(os.path.join(HERE, "liteflownet_64x96.npz"), **out)


def golden_monodepth2():
    from libs.deep_models.depth.monodepth2.resnet_encoder import ResnetEncoder
    from libs.deep_models.depth.monodepth2.depth_decoder import DepthDecoder
    enc = ResnetEncoder(18, False)
    dec = DepthDecoder(num_ch_enc=enc.num_ch_enc, scales=range(4))
    sd = O.monodepth2_state_dict(seed=4869)
    enc_sd = {k: v for k, v in sd.items() if k.startswith("encoder.")}
    # torchvision's resnet carries an unused fc layer + num_batches_tracked buffers: keep the module's own
    full = enc.state_dict()
    for k, v in enc_sd.items():
        assert k in full and full[k].shape == v.shape, k
        full[k] = v
    enc.load_state_dict(full)
    dec_sd = {k: v for k, v in sd.items() if k.startswith("decoder.")}
    print("DepthDecoder load_state_dict:", dec.load_state_dict(dec_sd, strict=True))
    enc.eval()
    dec.eval()
    g = torch.Generator().manual_seed(1002)
    img = F.avg_pool2d(torch.rand(1, 3, 64, 96, generator=g), 3, 1, 1)
    with torch.no_grad():
        feats = enc(img)
        outs = dec(feats)
    out = {"img": img.numpy()}
    for i, f in enumerate(feats):
        out["feat%d" % i] = f.numpy()
    for s in range(4):
        out["disp%d" % s] = outs[("disp", s)].numpy()
        print("  disp", s, tuple(outs[("disp", s)].shape))
    np.savez_compressed(os.path.join(HERE, "monodepth2_64x96.npz"), **out)


def kp_case(h, w, seed, frac):
    """seeded flow / flow_diff maps for the keypoint-selection fixtures (also imported by the tests)"""
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    diff = rng.random((int(h), int(w), 1), dtype=np.float32) * np.float32(0.1 / frac)
    diff[rng.random((int(h), int(w), 1)) < 0.01] = np.float32(0.05)  # ties
    flow = (rng.standard_normal((2, int(h), int(w))) * 3).astype(np.float32)
    return diff, flow


def golden_kp_selection():
    kps = load_by_path("ref_kp_selection", os.path.join(REF, "libs/matching/kp_selection.py"))
    from easydict import EasyDict
    cfg = EasyDict({"kp_selection": {
        "local_bestN": {"enable": True, "num_bestN": 2000, "num_row": 10, "num_col": 10, "score_method": "flow",
                        "thre": 0.1},
        "depth_consistency": {"enable": False, "thre": 0.05}}})
    out = {}
    for tag, (h, w, seed, frac) in {"a": (192, 640, 11, 0.6), "b": (376, 1241, 12, 0.35), "c": (100, 130, 13, 0.02),
                                    "d": (120, 160, 14, 0.004)}.items():
        diff, flow = kp_case(h, w, seed, frac)
        x = np.linspace(0, w - 1, w)
        y = np.linspace(0, h - 1, h)
        xv, yv = np.meshgrid(x, y)
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        res = kps.local_bestN(kp1=kp1, kp2=kp2, ref_data={"flow_diff": diff, "flow": flow}, cfg=cfg,
                              outputs={"good_kp_found": True})
        out[tag + "_spec"] = np.array([h, w, seed, frac])  # inputs are regenerated from the seed by kp_case()
        out[tag + "_good"] = np.array(res["good_kp_found"])
        if res["good_kp_found"]:
            out[tag + "_kp1"] = res["kp1_best"]
            out[tag + "_kp2"] = res["kp2_best"]
            print("  kp", tag, res["kp1_best"].shape)
        else:
            print("  kp", tag, "not enough keypoints")
    # score_method 'flow_ratio' (kp_selection.py:137-141,155): mask and score = flow_diff / |flow|; thresholds scaled so that
    # the ratio (diff ~ U[0, 0.1 / frac], |flow| ~ 3 sigma Rayleigh) leaves a mix of full and sparse cells
    cfg.kp_selection.local_bestN.score_method = "flow_ratio"
    for tag, (h, w, seed, frac, thre) in {"ra": (192, 640, 21, 0.6, 0.02), "rb": (376, 1241, 22, 0.35, 0.01),
                                          "rc": (100, 130, 23, 0.05, 0.02)}.items():
        diff, flow = kp_case(h, w, seed, frac)
        flow[:, 5:9, 7:30] = 0  # zero flow: the ratio is inf (nan where the map is 0 too) -- never selected
        diff[6, 10:14, 0] = 0
        cfg.kp_selection.local_bestN.thre = thre
        x = np.linspace(0, w - 1, w)
        y = np.linspace(0, h - 1, h)
        xv, yv = np.meshgrid(x, y)
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            res = kps.local_bestN(kp1=kp1, kp2=kp2, ref_data={"flow_diff": diff, "flow": flow}, cfg=cfg,
                                  outputs={"good_kp_found": True})
        out[tag + "_spec"] = np.array([h, w, seed, frac, thre])
        out[tag + "_good"] = np.array(res["good_kp_found"])
        if res["good_kp_found"]:
            out[tag + "_kp1"] = res["kp1_best"]
            out[tag + "_kp2"] = res["kp2_best"]
            print("  kp flow_ratio", tag, res["kp1_best"].shape)
        else:
            print("  kp flow_ratio", tag, "not enough keypoints")
    np.savez_compressed(os.path.join(HERE, "local_bestN.npz"), **out)


def kp_ratio_case(h, w, seed, frac):
    """inputs of the flow_ratio fixtures: kp_case + a patch of zero flow / zero consistency (also imported by the tests)"""
    diff, flow = kp_case(h, w, seed, frac)
    flow[:, 5:9, 7:30] = 0
    diff[6, 10:14, 0] = 0
    return diff, flow


def golden_kp_sampled():
    """the reference's sampled_kp + KeypointSampler.generate_kp_samples (ablation_correspondences_uniform.yml)"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    import libs.matching.kp_selection as kps
    import libs.matching.keypoint_sampler as sampler
    from easydict import EasyDict
    out = {}
    for tag, (h, w, seed, crop, nkp) in {"a": (192, 640, 41, [[0, 1], [0, 1]], 2000),
                                          "b": (376, 1241, 42, [[0.1, 0.9], [0.05, 0.95]], 2000),
                                          "c": (61, 83, 43, [[0.3, 1], [0, 0.7]], 37)}.items():
        _, flow = kp_case(h, w, seed, 0.5)
        cfg = EasyDict({"kp_selection": {"sampled_kp": {"enable": True, "num_kp": nkp}, "local_bestN": {"enable": False},
                                         "bestN": {"enable": False}},
                        "crop": {"flow_crop": crop}, "image": {"height": h, "width": w}})
        ks = sampler.KeypointSampler(cfg)
        x = np.linspace(0, w - 1, w)
        y = np.linspace(0, h - 1, h)
        xv, yv = np.meshgrid(x, y)
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        res = kps.sampled_kp(kp1=kp1, kp2=kp2, ref_data={"depth": np.zeros((h, w))}, kp_list=ks.kps["uniform"], cfg=cfg,
                             outputs={})
        out[tag + "_spec"] = np.array([h, w, seed, nkp] + [v for r in crop for v in r], np.float64)
        out[tag + "_idx"] = np.asarray(ks.kps["uniform"], np.int64)
        out[tag + "_kp1"] = res["kp1_list"]
        out[tag + "_kp2"] = res["kp2_list"]
        print("  sampled", tag, res["kp1_list"].shape)
    np.savez_compressed(os.path.join(HERE, "sampled_kp.npz"), **out)


BESTN_CASES = {"a": (192, 640, 71, 0.6, 2000, 0), "b": (376, 1241, 72, 0.35, 2000, 0), "c": (60, 90, 73, 0.5, 300, 0),
               "d": (120, 200, 74, 0.5, 1000, 1)}  # (h, w, seed, frac, N, with ties and NaNs)


def bestn_case(h, w, seed, frac, hard):
    diff, flow = kp_case(h, w, seed, frac)
    if hard:
        rng = np.random.Generator(np.random.PCG64(int(seed) + 1))
        diff = (np.round(diff * 400) / 400).astype(np.float32)       # heavy ties
        diff[rng.random(diff.shape) < 0.002] = np.float32("nan")      # NaN fails `>= 0`: dropped before the selection
    return diff, flow


def golden_kp_bestn():
    """the reference's bestN_flow_kp (ablation_correspondences_best_n.yml); needs the scalar numpy selection (re-exec)"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    import libs.matching.kp_selection as kps
    from easydict import EasyDict
    out = {}
    for tag, (h, w, seed, frac, N, hard) in BESTN_CASES.items():
        diff, flow = bestn_case(h, w, seed, frac, hard)
        cfg = EasyDict({"kp_selection": {"bestN": {"enable": True, "num_bestN": N}}})
        x = np.linspace(0, w - 1, w)
        y = np.linspace(0, h - 1, h)
        xv, yv = np.meshgrid(x, y)
        kp1 = np.expand_dims(np.transpose(np.stack([xv, yv]), (1, 2, 0)), 0)
        kp2 = kp1 + np.transpose(np.expand_dims(flow, 0), (0, 2, 3, 1))
        res = kps.bestN_flow_kp(kp1=kp1, kp2=kp2, ref_data={"flow_diff": diff}, cfg=cfg, outputs={})
        out[tag + "_kp1"] = res["kp1_best"]
        out[tag + "_kp2"] = res["kp2_best"]
        print("  bestN", tag, res["kp1_best"].shape)
    np.savez_compressed(os.path.join(HERE, "bestN.npz"), **out)


def golden_gric():
    gric = load_by_path("ref_gric", os.path.join(REF, "libs/tracker/gric.py"))
    rng = np.random.Generator(np.random.PCG64(21))
    n = 500
    kp1 = rng.random((n, 2)) * np.array([1241.0, 376.0])
    kp2 = kp1 + rng.standard_normal((n, 2)) * 2 + np.array([3.0, 0.5])
    Fm = rng.standard_normal((3, 3)) * np.array([[1e-6, 1e-5, 1e-3], [1e-5, 1e-6, 1e-2], [1e-3, 1e-2, 1.0]])
    Hm = np.eye(3) + rng.standard_normal((3, 3)) * np.array([[1e-3, 1e-3, 1.0], [1e-3, 1e-3, 1.0], [1e-6, 1e-6, 0]])
    f_res = gric.compute_fundamental_residual(Fm, kp1, kp2)
    h_res = gric.compute_homography_residual(Hm, kp1, kp2)
    np.savez_compressed(os.path.join(HERE, "gric.npz"), kp1=kp1, kp2=kp2, F=Fm, H=Hm, f_res=f_res, h_res=h_res,
                        f_gric=gric.calc_GRIC(f_res, 0.8, n, "EMat"), h_gric=gric.calc_GRIC(h_res, 0.8, n, "HMat"))
    print("  gric E", gric.calc_GRIC(f_res, 0.8, n, "EMat"), "H", gric.calc_GRIC(h_res, 0.8, n, "HMat"))


def golden_tracker():
    """the reference's own EssTracker (libs/tracker/E_tracker.py) over the oracle cv2 shim"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    import sklearn.linear_model as lm
    _RR = lm.RANSACRegressor

    def ransac_regressor_compat(base_estimator=None, **kw):  # sklearn 0.20 spelling: base_estimator=
        return _RR(estimator=base_estimator, **kw)
    lm.RANSACRegressor = ransac_regressor_compat
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = mpl.pyplot
    from easydict import EasyDict
    E_tracker = load_by_path("ref_E_tracker_pkg", os.path.join(REF, "libs/tracker/gric.py"))  # warm import path
    from libs.tracker.E_tracker import EssTracker
    from libs.general.timer import Timer
    from libs.geometry.camera_modules import Intrinsics, SE3
    cfg = EasyDict({
        "kp_selection": {"rigid_flow_kp": {"enable": False}},
        "e_tracker": {"ransac": {"reproj_thre": 0.2, "repeat": 5}, "validity": {"method": "GRIC", "thre": None},
                      "kp_src": "kp_best", "iterative_kp": {"enable": False}},
        "scale_recovery": {"method": "simple", "kp_src": "kp_best", "iterative_kp": {"enable": False, "kp_src": "kp_depth"},
                           "ransac": {"method": "depth_ratio", "min_samples": 3, "max_trials": 100, "stop_prob": 0.99,
                                      "thre": 0.1}},
        "image": {"height": 376, "width": 1241}})
    out = {}
    for tag, (seed, n, of, noise) in {"a": (31, 2000, 0.3, 0.15), "b": (32, 2000, 0.6, 0.3), "c": (33, 600, 0.2, 0.1),
                                      "d": (34, 2000, 0.97, 0.2)}.items():
        c = tracker_case(seed, n, of, noise)
        K = c["K"]
        cam = Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
        trk = EssTracker(cfg, cam, Timer())
        np.random.seed(4869 + seed)
        res = trk.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], True)
        pose = res["pose"]
        out[tag + "_spec"] = np.array([seed, n, of, noise])
        out[tag + "_pose"] = pose.pose.copy()
        out[tag + "_inliers"] = res["inliers"].copy()
        scale = -2.0
        if np.linalg.norm(pose.t) != 0:
            cur = {"kp_best": c["kp_cur"], "depth": c["depth_cur"]}
            ref = {"kp_best": c["kp_ref"]}
            scale = trk.scale_recovery(cur, ref, pose, False)["scale"]
        out[tag + "_scale"] = np.array(float(scale))
        st = np.random.get_state()
        out[tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
        print("  tracker", tag, "inliers", int(res["inliers"].sum()), "t", pose.t.ravel(), "scale", scale)
    np.savez_compressed(os.path.join(HERE, "e_tracker.npz"), **out)


from golden.make_golden_cases import PNP_CASES, pnp_case, tracker_case, variant_case  # noqa: E402,F401  (torch-free module: also imported by the second-environment run)


def golden_pnp_tracker():
    """the reference's own PnpTracker.compute_pose_3d2d (libs/tracker/pnp_tracker.py:45-125) over the oracle cv2 shim:
    the image / depth-range filters, 3 or 5 shuffled solvePnPRansac calls, best by inlier count, Rodrigues, pose inversion"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    from easydict import EasyDict
    from libs.tracker.pnp_tracker import PnpTracker
    from libs.geometry.camera_modules import Intrinsics
    cfg = EasyDict({"kp_selection": {"rigid_flow_kp": {"enable": False}}, "depth": {"max_depth": 50.0, "min_depth": 0.0},
                    "pnp_tracker": {"ransac": {"iter": 100, "reproj_thre": 1.0, "repeat": 5}},
                    "image": {"height": 376, "width": 1241}})
    out = {}
    for tag, (seed, n, of, noise, it, cop) in PNP_CASES.items():
        c = pnp_case(seed, n, of, noise, cop)
        K = c["K"]
        trk = PnpTracker(cfg, Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]]))
        np.random.seed(4869 + seed)
        res = trk.compute_pose_3d2d(c["kp1"], c["kp2"], c["depth_1"], it)
        out[tag + "_pose"] = res["pose"].pose.copy()
        out[tag + "_kp1"], out[tag + "_kp2"] = res["kp1"], res["kp2"]
        st = np.random.get_state()
        out[tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
        print("  pnp", tag, "survivors", len(res["kp1"]), "t", res["pose"].pose[:3, 3])
    np.savez_compressed(os.path.join(HERE, "pnp_tracker.npz"), **out)


def golden_tracker_flow():
    """the reference's EssTracker with e_tracker.validity.method 'flow' (ablation_model_sel_flow.yml) over the cv2 shim"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = mpl.pyplot
    from easydict import EasyDict
    from libs.tracker.E_tracker import EssTracker
    from libs.general.timer import Timer
    from libs.geometry.camera_modules import Intrinsics
    cfg = EasyDict({
        "kp_selection": {"rigid_flow_kp": {"enable": False}},
        "e_tracker": {"ransac": {"reproj_thre": 0.2, "repeat": 5}, "validity": {"method": "flow", "thre": 5},
                      "kp_src": "kp_best", "iterative_kp": {"enable": False}},
        "scale_recovery": {"method": "simple", "kp_src": "kp_best", "iterative_kp": {"enable": False, "kp_src": "kp_depth"},
                           "ransac": {"method": "depth_ratio", "min_samples": 3, "max_trials": 100, "stop_prob": 0.99,
                                      "thre": 0.1}},
        "image": {"height": 376, "width": 1241}})
    out = {}
    # a, b: ordinary motion (mean flow well above 5 px); c: nearly static pair (gate closed: identity pose and NO draw
    # from np.random); d: almost only outliers
    for tag, (seed, n, of, noise, shrink) in {"a": (51, 2000, 0.3, 0.15, 1.0), "b": (52, 1200, 0.6, 0.3, 1.0),
                                              "c": (53, 2000, 0.2, 0.1, 0.02), "d": (54, 2000, 0.97, 0.2, 1.0)}.items():
        c = tracker_case(seed, n, of, noise)
        kp_ref = c["kp_ref"]
        kp_cur = kp_ref + (c["kp_cur"] - kp_ref) * shrink
        K = c["K"]
        cam = Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
        trk = EssTracker(cfg, cam, Timer())
        np.random.seed(4869 + seed)
        res = trk.compute_pose_2d2d(kp_ref, kp_cur, True)
        out[tag + "_spec"] = np.array([seed, n, of, noise, shrink])
        out[tag + "_pose"] = res["pose"].pose.copy()
        out[tag + "_inliers"] = res["inliers"].copy()
        st = np.random.get_state()
        out[tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
        print("  tracker(flow)", tag, "inliers", int(res["inliers"].sum()), "t", res["pose"].t.ravel())
    np.savez_compressed(os.path.join(HERE, "e_tracker_flow.npz"), **out)


def golden_tracker_variants():
    """the reference's EssTracker with e_tracker.validity.method 'homo_ratio' and scale_recovery.ransac.method
    'abs_diff' (neither is in a shipped configuration; E_tracker.py:186-194,243-250,631-635) over the cv2 shim"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    import sklearn.linear_model as lm
    if not getattr(lm.RANSACRegressor, "_dfvo_compat", False):
        _RR = lm.RANSACRegressor

        def ransac_regressor_compat(base_estimator=None, **kw):  # sklearn 0.20 spelling: base_estimator=
            return _RR(estimator=base_estimator, **kw)
        ransac_regressor_compat._dfvo_compat = True
        lm.RANSACRegressor = ransac_regressor_compat
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = mpl.pyplot
    from easydict import EasyDict
    from libs.tracker.E_tracker import EssTracker
    from libs.general.timer import Timer
    from libs.geometry.camera_modules import Intrinsics
    cfg = EasyDict({
        "kp_selection": {"rigid_flow_kp": {"enable": False}},
        "e_tracker": {"ransac": {"reproj_thre": 0.2, "repeat": 5}, "validity": {"method": "homo_ratio", "thre": 0.4},
                      "kp_src": "kp_best", "iterative_kp": {"enable": False}},
        "scale_recovery": {"method": "simple", "kp_src": "kp_best", "iterative_kp": {"enable": False, "kp_src": "kp_depth"},
                           "ransac": {"method": "abs_diff", "min_samples": 3, "max_trials": 100, "stop_prob": 0.99,
                                      "thre": 0.1}},
        "image": {"height": 376, "width": 1241}})
    out = {}
    for tag in "abpd":
        c = variant_case(tag)
        K = c["K"]
        cam = Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
        trk = EssTracker(cfg, cam, Timer())
        np.random.seed(4869 + c["seed"])
        res = trk.compute_pose_2d2d(c["kp_ref"], c["kp_cur"], True)
        pose = res["pose"]
        out[tag + "_pose"] = pose.pose.copy()
        out[tag + "_inliers"] = res["inliers"].copy()
        scale = -2.0
        if np.linalg.norm(pose.t) != 0:
            scale = trk.scale_recovery({"kp_best": c["kp_cur"], "depth": c["depth_cur"]}, {"kp_best": c["kp_ref"]}, pose,
                                       False)["scale"]
        out[tag + "_scale"] = np.array(float(scale))
        st = np.random.get_state()
        out[tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
        print("  tracker(homo_ratio, abs_diff)", tag, "inliers", int(res["inliers"].sum()), "t", pose.t.ravel(), "scale", scale)
    np.savez_compressed(os.path.join(HERE, "e_tracker_variants.npz"), **out)


RIGID_CASES = {"a": (192, 640, 61, "opt_flow"), "b": (192, 640, 62, "rigid_flow"), "c": (120, 200, 63, "opt_flow")}


def rigid_case(h, w, seed):
    """inputs of the rigid-flow keypoint fixtures (also imported by the tests): a synthetic rigid scene, its reference
    depth as the float32 `raw_depth`, the true ref -> cur motion"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import rigid_scene
    sc = rigid_scene(int(h), int(w), seed=int(seed))
    T = np.eye(4)
    T[:3, :3] = sc["R"]
    T[:3, 3] = sc["t"]
    return dict(K=sc["K"], flow=sc["flow"], diff=sc["diff"], raw_depth=sc["depth_ref"].astype(np.float32),
                depth_cur=sc["depth_cur"], T_ref_to_cur=T)


def golden_rigid_flow():
    """the reference's EssTracker.kp_selection_good_depth / scale_recovery_iterative (RigidFlow layer + opt_rigid_flow_kp,
    ablation_scale_iterative.yml) on CPU torch"""
    import zlib
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    import sklearn.linear_model as lm
    if getattr(lm.RANSACRegressor, "__name__", "") != "ransac_regressor_compat":
        _RR = lm.RANSACRegressor

        def ransac_regressor_compat(base_estimator=None, **kw):
            return _RR(estimator=base_estimator, **kw)
        lm.RANSACRegressor = ransac_regressor_compat
    _to = torch.nn.Module.to
    torch.nn.Module.to = lambda self, *a, **k: self  # the layers are moved "to cuda" in their constructors
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = mpl.pyplot
    from easydict import EasyDict
    from libs.tracker.E_tracker import EssTracker
    from libs.general.timer import Timer
    from libs.geometry.camera_modules import Intrinsics, SE3
    out = {}
    for tag, (h, w, seed, score) in RIGID_CASES.items():
        c = rigid_case(h, w, seed)
        K = c["K"]
        cfg = EasyDict({
            "kp_selection": {"rigid_flow_kp": {"enable": True, "num_bestN": 2000, "num_row": 10, "num_col": 10,
                                               "score_method": score, "rigid_flow_thre": 5, "optical_flow_thre": 0.1}},
            "e_tracker": {"ransac": {"reproj_thre": 0.2, "repeat": 5}, "validity": {"method": "GRIC", "thre": None},
                          "kp_src": "kp_best", "iterative_kp": {"enable": False, "kp_src": "kp_depth", "score_method": score}},
            "scale_recovery": {"method": "iterative", "kp_src": "kp_depth",
                               "iterative_kp": {"enable": False, "kp_src": "kp_depth", "score_method": score},
                               "ransac": {"method": "depth_ratio", "min_samples": 3, "max_trials": 100, "stop_prob": 0.99,
                                          "thre": 0.1}},
            "image": {"height": h, "width": w}})
        cam = Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
        trk = EssTracker(cfg, cam, Timer())
        ref = {"flow": c["flow"], "flow_diff": c["diff"][..., None], "raw_depth": c["raw_depth"],
               "rigid_flow_pose": SE3(c["T_ref_to_cur"])}
        cur = {"depth": c["depth_cur"]}
        res = trk.kp_selection_good_depth(cur, ref, score)
        out[tag + "_spec"] = np.array([h, w, seed, 0 if score == "opt_flow" else 1])
        for k in ("kp1_depth", "kp2_depth", "kp1_depth_uniform", "kp2_depth_uniform"):
            out[tag + "_" + k] = res[k]
        m = np.ascontiguousarray(res["rigid_flow_mask"], np.float32)
        out[tag + "_mask_crc"] = np.array(zlib.crc32(m.tobytes()), np.uint32)
        out[tag + "_mask_rows"] = m[:: max(1, h // 8)].copy()
        # scale_recovery_iterative from prev_scale = 0 with the unit-translation pose of the E-tracker (cur -> ref)
        E_pose = SE3(np.linalg.inv(c["T_ref_to_cur"]))
        E_pose.t = E_pose.t / np.linalg.norm(E_pose.t)
        np.random.seed(4869 + seed)
        trk.prev_scale = 0
        it = trk.scale_recovery_iterative(cur, ref, E_pose)
        out[tag + "_iter_scale"] = np.array(float(it["scale"]))
        out[tag + "_iter_cur_kp"] = it["cur_kp"]
        out[tag + "_iter_ref_kp"] = it["ref_kp"]
        st = np.random.get_state()
        out[tag + "_rng_after"] = np.r_[st[1].astype(np.uint32), np.uint32(st[2])]
        print("  rigid", tag, res["kp1_depth"].shape, res["kp1_depth_uniform"].shape, "iterative scale", it["scale"])
    torch.nn.Module.to = _to
    np.savez_compressed(os.path.join(HERE, "rigid_flow_kp.npz"), **out)


LANCZOS_CASES = [  # (seed, H, W, out_h, out_w): KITTI frame -> monodepth2 feed, RobotCar crop -> feed, down / up / one axis
    (31, 376, 1241, 192, 640), (32, 768, 1280, 256, 640), (33, 37, 53, 20, 31), (34, 20, 31, 37, 53),
    (35, 100, 100, 100, 57), (36, 64, 48, 192, 48)]


def lanczos_case(seed, h, w):
    """seeded uint8 test image: smooth structure + noise + saturated patches (also imported by the tests)"""
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    yy, xx = np.mgrid[0:h, 0:w]
    base = 128 + 90 * np.sin(xx / 7.0)[..., None] * np.cos(yy / 5.0)[..., None] * np.array([1.0, 0.7, -0.8])
    img = np.clip(base + rng.normal(0, 40, (h, w, 3)), 0, 255).astype(np.uint8)
    img[: h // 5, : w // 4] = 255
    img[-(h // 6):, -(w // 3):] = 0
    return img


def golden_lanczos():
    """Pillow itself (the reference's forward_depth calls img.resize(..., pil.LANCZOS), deep_models.py:195-199)"""
    import zlib
    from PIL import Image
    import PIL
    out = {"pillow_version": np.array(PIL.__version__)}
    for seed, h, w, oh, ow in LANCZOS_CASES:
        img = lanczos_case(seed, h, w)
        res = np.asarray(Image.fromarray(img).resize((ow, oh), Image.LANCZOS))
        key = "%d_%dx%d_%dx%d" % (seed, h, w, oh, ow)
        out["crc_" + key] = np.array(zlib.crc32(res.tobytes()), np.uint32)
        out["rows_" + key] = res[:: max(1, oh // 8)].copy()  # every (oh/8)-th output row in full
    np.savez_compressed(os.path.join(HERE, "lanczos.npz"), **out)
    print("wrote lanczos.npz (Pillow %s)" % PIL.__version__)


def _stub_matplotlib():
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = mpl.pyplot


def kitti_eval_reference(gt, res):
    """the reference's own KittiEvalOdom methods (tools/evaluation/odometry/kitti_odometry.py:190-245,274-299,445-497 and
    the reporting arithmetic of :655-683) on two pose lists"""
    _stub_matplotlib()
    ko = load_by_path("ref_kitti_odometry", os.path.join(REF, "tools/evaluation/odometry/kitti_odometry.py"))
    ev = ko.KittiEvalOdom()
    pg = {i: np.array(p, dtype=np.float64) for i, p in enumerate(gt)}
    pr = {i: np.array(p, dtype=np.float64) for i, p in enumerate(res)}
    seq_err = ev.calc_sequence_errors(pg, pr)
    ave_t, ave_r = ev.compute_overall_err(seq_err)
    rpe = ev.compute_RPE(pg, pr)
    return {"seq_err": np.array(seq_err), "t_rel": ave_t * 100, "r_rel": ave_r / np.pi * 180 * 100, "ate": ev.compute_ATE(pg, pr),
            "rpe_t": float(np.mean(np.asarray(rpe["trans"]))), "rpe_r": float(np.mean(np.asarray(rpe["rot"])) * 180 / np.pi)}


def golden_kitti_eval():
    """kitti_eval.npz: the reference evaluator on (rendered ground truth, oracle trajectory) of tunnel_traj.npz -- pins
    oracle/kitti_eval.py (tests/test_oracle_eval.py)"""
    fx = np.load(os.path.join(HERE, "tunnel_traj.npz"))
    gt, res = fx["gt"], fx["poses"]
    r = kitti_eval_reference(list(gt), list(res))
    print("  reference evaluator: t_rel %.4f %%, r_rel %.4f deg/100m, ATE %.3f, RPE %.4f m %.4f deg, %d segments" % (
        r["t_rel"], r["r_rel"], r["ate"], r["rpe_t"], r["rpe_r"], len(r["seq_err"])))
    np.savez_compressed(os.path.join(HERE, "kitti_eval.npz"), gt=gt, res=res, **r)


EVAL_ALIGNMENTS = [None, "scale", "scale_7dof", "7dof", "6dof"]


def eval_align_cases():
    """Trajectories for the alignment modes: (a) the committed tunnel trajectory; (b) a 700-frame KITTI-like drive (~0.9 m
    per frame, gentle turns) whose estimate drifts in scale and heading and whose FIRST poses are not the identity (so that
    eval()'s first-frame normalisation matters); (c) the same estimate cut to 430 frames (result shorter than ground truth)."""
    fx = np.load(os.path.join(HERE, "tunnel_traj.npz"))
    cases = {"tunnel": (fx["gt"], fx["poses"])}
    rng = np.random.Generator(np.random.PCG64(77))

    def rot(ax, a):
        c, s_ = np.cos(a), np.sin(a)
        R = np.eye(3)
        i, j = [(1, 2), (0, 2), (0, 1)][ax]
        R[i, i], R[i, j], R[j, i], R[j, j] = c, -s_, s_, c
        return R

    def drive(n, yaw_rate, scale, noise):
        T = np.eye(4)
        T[:3, :3] = rot(1, 0.3) @ rot(0, -0.05)
        T[:3, 3] = [12.0, -1.5, 40.0]
        out = [T.copy()]
        for k in range(1, n):
            d = np.eye(4)
            d[:3, :3] = rot(1, yaw_rate(k)) @ rot(0, noise * rng.normal()) @ rot(2, noise * rng.normal())
            d[:3, 3] = np.array([0.01 * np.sin(k / 40.0), -0.004, 0.9]) * scale(k) + noise * rng.normal(size=3)
            T = T @ d
            out.append(T.copy())
        return np.array(out)

    gt = drive(700, lambda k: 0.004 * np.sin(k / 90.0), lambda k: 1.0, 0.0)
    est = drive(700, lambda k: 0.004 * np.sin(k / 90.0) + 2e-5, lambda k: 0.93 + 1e-4 * k, 2e-3)
    A = np.eye(4)  # the estimate lives in another world frame
    A[:3, :3] = rot(1, -0.8) @ rot(2, 0.1)
    A[:3, 3] = [-3.0, 0.7, 5.0]
    est = A @ est
    cases["drive"] = (gt, est)
    cases["drive_short"] = (gt, est[:430])
    return cases


def kitti_eval_reference_eval(gt, res, alignment):
    """KittiEvalOdom.eval itself (kitti_odometry.py:556-700) on one sequence written to a temporary directory; plotting and
    the error files are switched off, write_result hands over the five numbers at full precision"""
    import tempfile
    _stub_matplotlib()
    import matplotlib
    matplotlib.use("Agg")
    ko = load_by_path("ref_kitti_odometry", os.path.join(REF, "tools/evaluation/odometry/kitti_odometry.py"))
    got = {}

    class Ev(ko.KittiEvalOdom):
        def plot_trajectory(self, *a, **k):
            pass

        def plot_error(self, *a, **k):
            pass

        def write_result(self, f, seq, errs):
            got["errs"] = [float(e) for e in errs]

    def write(path, poses):
        with open(path, "w") as f:
            for i, p in enumerate(poses):
                f.write(str(i) + " " + " ".join(repr(float(v)) for v in np.asarray(p).flatten()[:12]) + "\n")

    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "gt"))
        os.makedirs(os.path.join(d, "res"))
        write(os.path.join(d, "gt", "09.txt"), gt)
        write(os.path.join(d, "res", "09.txt"), res)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            Ev().eval(os.path.join(d, "gt"), os.path.join(d, "res"), alignment=alignment, seqs=["09"])
    t, r, ate, rt, rr = got["errs"]
    return {"t_rel": t * 100, "r_rel": r / np.pi * 180 * 100, "ate": ate, "rpe_t": rt, "rpe_r": rr * 180 / np.pi}


def golden_kitti_eval_align():
    """kitti_eval_align.npz: KittiEvalOdom.eval under every alignment mode -- pins oracle/kitti_eval.py's align() and,
    through it, df-vo_amd/evaluation.py (tests/test_oracle_eval.py, tests/test_evaluation_cpu.py)"""
    out = {}
    for name, (gt, res) in eval_align_cases().items():
        out[name + "_gt"], out[name + "_res"] = gt, res
        for al in EVAL_ALIGNMENTS:
            r = kitti_eval_reference_eval(gt, res, al)
            out["%s_%s" % (name, al)] = np.array([r[k] for k in ("t_rel", "r_rel", "ate", "rpe_t", "rpe_r")])
            print("  %-12s %-10s t_rel %.4f %%  r_rel %.4f  ATE %.4f  RPE %.5f m %.5f deg" % (
                name, al, r["t_rel"], r["r_rel"], r["ate"], r["rpe_t"], r["rpe_r"]))
    np.savez_compressed(os.path.join(HERE, "kitti_eval_align.npz"), **out)


TARGET_SIZE_CASES = [(192, 640), (376, 1241), (370, 1226), (375, 1242), (384, 1248), (256, 640), (128, 416), (70, 100), (64, 96),
                     (960, 1280), (1280, 1920), (480, 640), (320, 1024), (256, 832), (200, 300), (97, 301), (1000, 1000)]


def golden_target_size():
    """target_size.npz: the reference's own DeepFlow.get_target_size (deep_flow.py:89-105) -- whose rebinding of h, w makes
    it return (floor, floor) multiples of 32, or (ceil, ceil) when float rounding leaves entry [0][0] non-zero"""
    from oracle import cv2_shim
    sys.modules["cv2"] = cv2_shim
    from libs.deep_models.flow.deep_flow import DeepFlow
    df = DeepFlow.__new__(DeepFlow)
    out = np.array([[h, w] + [int(v) for v in df.get_target_size(h, w)] for h, w in TARGET_SIZE_CASES])
    for r in out:
        print("  %4d x %4d -> %4d x %4d" % tuple(r))
    np.savez_compressed(os.path.join(HERE, "target_size.npz"), cases=out)


def golden_dfvo_main(n_frames=5, h=256, w=640):
    """dfvo_main.npz: the reference's OWN frame loop -- apis/run.py:76-92 + DFVO.main / deep_model_inference / tracking /
    update_global_pose (libs/dfvo.py:347-425,299-345,121-262,109-119), its DeepModel / LiteFlow / Monodepth2DepthNet /
    KeypointSampler / EssTracker / PnpTracker classes, all unmodified, over the third-party shims -- on a coded tunnel
    sequence, with the crafted weights loaded from files in the reference's on-disk formats through a stub Dataset.
    Expected values of tests/test_dropin_gpu.py::test_mirrors_reproduce_reference_main_loop and of
    tests/test_oracle_pipeline.py (which pins oracle/pipeline_np.py to this loop)."""
    import tempfile
    from oracle import cv2_shim
    cv2_shim.imwrite = lambda *a, **k: True
    sys.modules["cv2"] = cv2_shim
    _stub_matplotlib()
    import sklearn.linear_model as lm
    if not getattr(lm.RANSACRegressor, "_dfvo_compat", False):
        _RR = lm.RANSACRegressor

        def ransac_regressor_compat(base_estimator=None, **kw):  # sklearn 0.20 spelling
            return _RR(estimator=base_estimator, **kw)
        ransac_regressor_compat._dfvo_compat = True
        lm.RANSACRegressor = ransac_regressor_compat
    from libs.deep_models.flow.lite_flow_net import correlation as ref_corr
    ref_corr.FunctionCorrelation = lambda tensorFirst, tensorSecond, intStride: O.correlation(
        tensorFirst, tensorSecond, intStride)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, write_weight_files
    seq = coded_tunnel_sequence(h, w, n_frames, mode="mux", step=1.0, seed=31)
    tmp = tempfile.mkdtemp(prefix="dfvo_golden_")
    flow_path, depth_dir = write_weight_files(tmp, crafted_liteflownet_state_dict(h, w, "mux"), crafted_monodepth2_state_dict())
    # configuration exactly as apis/run.py builds it, from the reference's own default file
    from libs.general.configuration import ConfigLoader
    cfg = ConfigLoader().merge_cfg([os.path.join(REF, "options/examples/default_configuration.yml")])
    cfg.dataset = "coded_tunnel"
    cfg.seq = "00"
    cfg.image.height, cfg.image.width = h, w
    cfg.directory.result_dir = os.path.join(tmp, "result")
    cfg.directory.gt_pose_dir = None
    cfg.depth.deep_depth.pretrained_model = depth_dir
    cfg.deep_flow.flow_net_weight = flow_path
    cfg.visualization.enable = False
    cfg.no_confirm = True
    os.makedirs(cfg.directory.result_dir, exist_ok=True)
    from libs.geometry.camera_modules import Intrinsics
    from libs import datasets as ref_datasets
    K = seq["K"]

    class CodedTunnel:
        """stub of libs/datasets/dataset.py's interface as DFVO uses it"""

        def __init__(self, cfg):
            self.cfg = cfg
            self.cam_intrinsics = Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
            self.data_dir = {"depth_src": None}
            self.gt_poses = {i: p for i, p in enumerate(seq["poses"])}

        def __len__(self):
            return n_frames

        def get_timestamp(self, img_id):
            return img_id

        def get_image(self, timestamp):
            return seq["frames"][timestamp].copy()

        def save_result_traj(self, traj_txt, poses):
            from libs.general.utils import save_traj
            save_traj(traj_txt, {i: poses[i].pose for i in poses}, format="kitti")
    ref_datasets.datasets["coded_tunnel"] = CodedTunnel
    from libs.dfvo import DFVO
    np.random.seed(cfg.seed)   # apis/run.py:81-84
    torch.manual_seed(cfg.seed)
    vo = DFVO(cfg)
    modes, kps = [], []
    _tracking = vo.tracking

    def tracking_and_record():
        _tracking()
        modes.append(vo.tracking_mode)
        kps.append(len(vo.ref_data["kp_best"]) if vo.tracking_stage >= 1 and "kp_best" in vo.ref_data else 0)
    vo.tracking = tracking_and_record
    vo.main()
    poses = np.stack([vo.global_poses[i].pose for i in range(n_frames)])
    st = np.random.get_state()
    print("  reference DFVO.main:", modes, "kp", kps, "final t", poses[-1][:3, 3])
    traj = open(os.path.join(cfg.directory.result_dir, "00.txt")).read()
    np.savez_compressed(os.path.join(HERE, "dfvo_main.npz"), poses=poses, modes=np.array(modes), n_kp=np.array(kps),
                        rng_after=np.r_[st[1].astype(np.uint32), np.uint32(st[2])], gt=seq["poses"], K=K,
                        n_frames=n_frames, h=h, w=w, seq_seed=31, traj_txt=np.array(traj))


if __name__ == "__main__":
    apply_compat()
    torch.set_num_threads(8)
    which = [a for a in sys.argv[1:] if not a.startswith("--")]
    todo = {"liteflownet": golden_liteflownet, "monodepth2": golden_monodepth2, "kp": golden_kp_selection,
            "gric": golden_gric, "tracker": golden_tracker, "lanczos": golden_lanczos, "sampled": golden_kp_sampled,
            "tracker_flow": golden_tracker_flow, "rigid": golden_rigid_flow, "bestn": golden_kp_bestn,
            "kitti_eval": golden_kitti_eval, "kitti_eval_align": golden_kitti_eval_align, "dfvo_main": golden_dfvo_main,
            "target_size": golden_target_size, "tracker_variants": golden_tracker_variants, "pnp_tracker": golden_pnp_tracker}
    for name, fn in todo.items():
        if not which or name in which:
            print("==", name)
            fn()
