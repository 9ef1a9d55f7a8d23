"""GPU: the drop-in boundary as libs/dfvo.py uses it.  `track_sequence` below restates the control flow of
DFVO.tracking (/root/reference/libs/dfvo.py:121-262: keypoint selection, E-tracker, scale recovery, optional iterative
refinement, PnP fallback, update_global_pose :109-119) against an object with the reference's class surface.  It is run
once over the mirror classes (df-vo_amd/libs/**, everything on the device) and once over the same surface backed by the
oracle, on a short synthetic sequence with the numpy RandomState carried from pair to pair: poses, tracking modes and
the RandomState must agree bit for bit (pose tolerance of the contract: 1e-4 Frobenius)."""
import copy
import importlib

import numpy as np
import pytest

from oracle import tracker_np as T
from synth import rigid_scene

pytestmark = pytest.mark.gpu


class NS(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def make_cfg(h, w, selector="local_bestN", validity="GRIC", scale_method="simple", kp_score="flow", ransac_method="depth_ratio"):
    kp_src = "kp_list" if selector == "sampled_kp" else "kp_best"
    it = NS(enable=False, kp_src="kp_depth", score_method="opt_flow")
    return NS(
        image=NS(height=h, width=w), crop=NS(flow_crop=[[0, 1], [0, 1]]), tracking_method="hybrid",
        depth=NS(min_depth=0.0, max_depth=50.0),
        kp_selection=NS(local_bestN=NS(enable=selector == "local_bestN", num_bestN=2000, num_row=10, num_col=10,
                                       score_method=kp_score, thre=0.1),
                        bestN=NS(enable=selector == "bestN", num_bestN=2000),
                        sampled_kp=NS(enable=selector == "sampled_kp", num_kp=2000),
                        rigid_flow_kp=NS(enable=scale_method == "iterative", num_bestN=2000, num_row=10, num_col=10,
                                         score_method="opt_flow", rigid_flow_thre=5, optical_flow_thre=0.1),
                        depth_consistency=NS(enable=False, thre=0.05)),
        e_tracker=NS(ransac=NS(reproj_thre=0.2, repeat=5), validity=NS(method=validity, thre={"flow": 5, "homo_ratio": 0.6}.get(validity)),
                     kp_src=kp_src, iterative_kp=NS(it)),
        scale_recovery=NS(method=scale_method, kp_src="kp_depth" if scale_method == "iterative" else kp_src,
                          iterative_kp=NS(it),
                          ransac=NS(method=ransac_method, min_samples=3, max_trials=100, stop_prob=0.99, thre=0.1)),
        pnp_tracker=NS(ransac=NS(iter=100, reproj_thre=1.0, repeat=5), kp_src=kp_src, iterative_kp=NS(it)))


class OracleSurface:
    """KeypointSampler / EssTracker / PnpTracker methods as track_sequence calls them, answered by oracle/tracker_np.py"""

    def __init__(self, cfg, K, SE3):
        self.cfg, self.K, self.SE3, self.prev_scale = cfg, K, SE3, 0
        if cfg.kp_selection.sampled_kp.enable:
            self.idx = T.generate_kp_samples(cfg.image.height, cfg.image.width, cfg.crop.flow_crop, cfg.kp_selection.sampled_kp.num_kp)

    def kp_selection(self, cur, ref):
        ks = self.cfg.kp_selection
        out = {"good_kp_found": True}
        if ks.local_bestN.enable:
            r = T.local_bestN(ref["flow"], ref["flow_diff"], score_method=ks.local_bestN.score_method)
            out["good_kp_found"] = r["good_kp_found"]
            if r["good_kp_found"]:
                out["kp1_best"], out["kp2_best"] = r["kp1_best"], r["kp2_best"]
        elif ks.bestN.enable:
            out["kp1_best"], out["kp2_best"] = T.bestN_flow_kp(ref["flow"], ref["flow_diff"], ks.bestN.num_bestN)
        if ks.sampled_kp.enable:
            out["kp1_list"], out["kp2_list"] = T.sampled_kp(ref["flow"], self.idx, self.cfg.crop.flow_crop)
        return out

    def update_kp_data(self, cur, ref, out):
        if "kp1_best" in out:
            ref["kp_best"], cur["kp_best"] = out["kp1_best"][0], out["kp2_best"][0]
        if "kp1_list" in out:
            ref["kp_list"], cur["kp_list"] = out["kp1_list"][0], out["kp2_list"][0]

    def compute_pose_2d2d(self, kp_ref, kp_cur, is_iterative):
        v = self.cfg.e_tracker.validity
        r = T.compute_pose_2d2d(kp_ref, kp_cur, self.K, validity=v.method, validity_thre=v.thre)
        pose = self.SE3()
        pose.R, pose.t = r["R"], r["t"]
        return {"pose": pose, "inliers": r["inliers"]}

    def scale_recovery(self, cur, ref, E_pose, is_iterative):
        if self.cfg.scale_recovery.method == "simple":
            src = self.cfg.scale_recovery.kp_src
            return {"scale": T.find_scale_from_depth(ref[src], cur[src], E_pose.inv_pose, cur["depth"], self.K,
                                                     method=self.cfg.scale_recovery.ransac.method)}
        r = T.scale_recovery_iterative(ref["flow"], ref["flow_diff"], ref["raw_depth"], cur["depth"], E_pose.pose, self.K,
                                       self.prev_scale, self.cfg.scale_recovery.iterative_kp.score_method)
        self.prev_scale = r["scale"]
        return {"scale": r["scale"], "cur_kp_depth": r["cur_kp"], "ref_kp_depth": r["ref_kp"],
                "rigid_flow_mask": r["rigid_flow_mask"]}

    def compute_pose_3d2d(self, kp1, kp2, depth_1, is_iterative):
        r = T.compute_pose_3d2d(kp1, kp2, depth_1, self.K, 0.0, 50.0, 5, 100, 1.0)
        return {"pose": self.SE3(r["pose"])}


def track_sequence(frames, cfg, sampler, e_tracker, pnp_tracker, SE3):
    """dfvo.py:121-262 for tracking_method 'hybrid' without the iterative keypoint refinement; frames[k] holds what
    deep_model_inference leaves in ref_data / cur_data for pair k"""
    global_pose = SE3()
    out = []
    for fr in frames:
        ref = {"flow": fr["flow"], "flow_diff": fr["diff"][..., None], "depth": fr["depth_ref"],
               "raw_depth": fr["depth_ref"].astype(np.float32)}
        cur = {"depth": fr["depth_cur"]}
        kp_sel = sampler.kp_selection(cur, ref)                                  # dfvo.py:147
        if not kp_sel["good_kp_found"]:
            out.append(("Constant motion", global_pose.pose.copy()))              # dfvo.py:151-161
            continue
        sampler.update_kp_data(cur, ref, kp_sel)
        hybrid_pose = SE3()
        mode = "Ess. Mat."
        e_out = e_tracker.compute_pose_2d2d(ref[cfg.e_tracker.kp_src], cur[cfg.e_tracker.kp_src], True)   # :168
        E_pose = e_out["pose"]
        hybrid_pose.R = E_pose.R
        scale = -1
        if np.linalg.norm(E_pose.t) != 0:                                        # :184
            scale = e_tracker.scale_recovery(cur, ref, E_pose, False)["scale"]
            if scale != -1:
                hybrid_pose.t = E_pose.t * scale
        if np.linalg.norm(E_pose.t) == 0 or scale == -1:                         # :227
            p = pnp_tracker.compute_pose_3d2d(ref[cfg.pnp_tracker.kp_src], cur[cfg.pnp_tracker.kp_src], ref["depth"], True)
            hybrid_pose = p["pose"]
            mode = "PnP"
        pose = copy.deepcopy(hybrid_pose)
        global_pose.t = global_pose.R @ pose.t * 1 + global_pose.t                # update_global_pose, dfvo.py:109-119
        global_pose.R = global_pose.R @ pose.R
        out.append((mode, global_pose.pose.copy()))
    return out


@pytest.mark.parametrize("selector,validity,scale_method", [("local_bestN", "GRIC", "simple"), ("sampled_kp", "flow", "simple"),
                                                            ("bestN", "GRIC", "simple"), ("local_bestN", "GRIC", "iterative"),
                                                            ("local_bestN", "flow", "iterative"),
                                                            ("local_bestN:flow_ratio", "homo_ratio", "simple:abs_diff")])
def test_tracking_loop_over_mirrors_equals_oracle(gpu, selector, validity, scale_method):
    """the last case runs the three branches no shipped configuration selects: local_bestN.score_method 'flow_ratio',
    validity.method 'homo_ratio', scale_recovery.ransac.method 'abs_diff'"""
    h, w = 192, 640
    selector, _, kp_score = selector.partition(":")
    scale_method, _, ransac_method = scale_method.partition(":")
    cfg = make_cfg(h, w, selector, validity, scale_method, kp_score or "flow", ransac_method or "depth_ratio")
    frames = [rigid_scene(h, w, seed=300 + i, bad_frac=0.3 + 0.1 * i) for i in range(3)]
    K = frames[0]["K"]
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    ks_mod = importlib.import_module("df-vo_amd.libs.matching.keypoint_sampler")
    trk_mod = importlib.import_module("df-vo_amd.libs.tracker")
    cam = cam_mod.Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
    np.random.seed(4869)
    got = track_sequence(frames, cfg, ks_mod.KeypointSampler(cfg), trk_mod.EssTracker(cfg, cam, None),
                         trk_mod.PnpTracker(cfg, cam), cam_mod.SE3)
    st_hip = np.random.get_state()
    np.random.seed(4869)
    ora = OracleSurface(cfg, K, cam_mod.SE3)
    want = track_sequence(frames, cfg, ora, ora, ora, cam_mod.SE3)
    st_ref = np.random.get_state()
    print("modes:", [m for m, _ in got], "| final translation", got[-1][1][:3, 3])
    assert [m for m, _ in got] == [m for m, _ in want]
    for (_, a), (_, b) in zip(got, want):
        assert np.abs(a - b).max() <= 1e-12, np.abs(a - b).max()
    assert np.array_equal(st_hip[1], st_ref[1]) and st_hip[2] == st_ref[2]
    assert any(np.linalg.norm(p[:3, 3]) > 0.3 for _, p in got)  # the sequence really moved


# -----------------------------------------------------------------------------------------------------------------------
# The mirrors instantiated the way libs/dfvo.py instantiates them (DeepModel(cfg).initialize_models() from weight FILES in the
# reference's on-disk formats, KeypointSampler(cfg), EssTracker / PnpTracker(cfg, cam_intrinsics)) and driven through the
# frame loop of DFVO.main; the expected poses / tracking modes come from the REFERENCE's own DFVO.main() on the same frames
# and weight files (tests/golden/make_golden.py dfvo_main -> dfvo_main.npz; the reference cannot travel to the GPU box).
# -----------------------------------------------------------------------------------------------------------------------
def full_cfg(h, w, flow_path, depth_dir):
    """the keys of options/examples/default_configuration.yml that the hot path reads"""
    c = make_cfg(h, w)
    c["dataset"] = "kitti_odom"
    c["seed"] = 4869
    c["depth"] = NS(depth_src=None, min_depth=0.0, max_depth=50.0,
                    deep_depth=NS(network="monodepth2", pretrained_model=depth_dir))
    c["deep_flow"] = NS(network="liteflow", flow_net_weight=flow_path, forward_backward=True)
    c["deep_pose"] = NS(enable=False)
    c["online_finetune"] = NS(enable=False, flow=NS(enable=True), depth=NS(enable=False))
    c["crop"] = NS(depth_crop=[[0.3, 1], [0, 1]], flow_crop=[[0, 1], [0, 1]])
    return c


def _build_mirrors(cfg, K):
    dm_mod = importlib.import_module("df-vo_amd.libs.deep_models.deep_models")
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    ks_mod = importlib.import_module("df-vo_amd.libs.matching.keypoint_sampler")
    trk_mod = importlib.import_module("df-vo_amd.libs.tracker")
    deep_models = dm_mod.DeepModel(cfg)                     # dfvo.py:78-79
    deep_models.initialize_models()
    cam = cam_mod.Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
    sampler = ks_mod.KeypointSampler(cfg)                   # dfvo.py:75
    e_tracker, pnp_tracker = trk_mod.EssTracker(cfg, cam, None), trk_mod.PnpTracker(cfg, cam)   # dfvo.py:100-103
    return deep_models, sampler, e_tracker, pnp_tracker, cam_mod.SE3


def _main_loop(cfg, seq, n, h, w, mirrors, record=None):
    """the frame loop of DFVO.main (dfvo.py:358-404) over the mirror classes; record (optional list) receives, per pair, what
    the calls returned (for the session-on / session-off comparison)"""
    from oracle import cv2_shim
    deep_models, sampler, e_tracker, pnp_tracker, SE3 = mirrors
    np.random.seed(cfg.seed)                                # apis/run.py:81-84
    ref_data, cur_data = {}, {}
    global_pose = SE3()
    poses, modes = [], []
    for img_id in range(n):
        cur_data["id"], cur_data["timestamp"], cur_data["img"] = img_id, img_id, seq["frames"][img_id].copy()
        # deep_model_inference, dfvo.py:299-345
        raw = deep_models.forward_depth(imgs=[cur_data["img"]])
        assert raw.dtype == np.float32 and raw.shape == (192, 640)
        cur_data["raw_depth"] = cv2_shim.resize(raw, (w, h), interpolation=cv2_shim.INTER_NEAREST)
        cur_data["depth"] = T.preprocess_depth(cur_data["raw_depth"], cfg.crop.depth_crop, [cfg.depth.min_depth, cfg.depth.max_depth])
        mode = "Ess. Mat."
        if img_id >= 1:
            flows = deep_models.forward_flow(cur_data, ref_data, forward_backward=cfg.deep_flow.forward_backward)
            kf, kb, kd = (ref_data["id"], cur_data["id"]), (cur_data["id"], ref_data["id"]), (ref_data["id"], cur_data["id"], "diff")
            assert set(flows) == {kf, kb, kd}
            assert flows[kf].shape == (2, h, w) and flows[kb].shape == (2, h, w) and flows[kd].shape == (h, w, 1)
            assert all(v.dtype == np.float32 for v in flows.values())
            ref_data["flow"], cur_data["flow"], ref_data["flow_diff"] = flows[kf].copy(), flows[kb].copy(), flows[kd].copy()
            # tracking, dfvo.py:139-262
            kp_sel = sampler.kp_selection(cur_data, ref_data)
            assert kp_sel["good_kp_found"]
            sampler.update_kp_data(cur_data, ref_data, kp_sel)
            hybrid = SE3()
            e_out = e_tracker.compute_pose_2d2d(ref_data["kp_best"], cur_data["kp_best"], True)
            E_pose = e_out["pose"]
            hybrid.R = E_pose.R
            scale = -1
            if np.linalg.norm(E_pose.t) != 0:
                scale = e_tracker.scale_recovery(cur_data, ref_data, E_pose, False)["scale"]
                if scale != -1:
                    hybrid.t = E_pose.t * scale
            if np.linalg.norm(E_pose.t) == 0 or scale == -1:
                hybrid = pnp_tracker.compute_pose_3d2d(ref_data["kp_best"], cur_data["kp_best"], ref_data["depth"], True)["pose"]
                mode = "PnP"
            global_pose.t = global_pose.R @ hybrid.t + global_pose.t      # update_global_pose, dfvo.py:109-119
            global_pose.R = global_pose.R @ hybrid.R
            if record is not None:
                record.append({"raw": np.array(raw), "fwd": np.array(ref_data["flow"]), "bwd": np.array(cur_data["flow"]),
                               "diff": np.array(ref_data["flow_diff"]), "kp_ref": np.array(ref_data["kp_best"]),
                               "kp_cur": np.array(cur_data["kp_best"]), "inliers": np.array(e_out["inliers"]),
                               "R": np.array(E_pose.R), "t": np.array(E_pose.t), "scale": scale,
                               "rng": np.random.get_state()[1].copy()})
        poses.append(global_pose.pose.copy())
        modes.append(mode)
        ref_data = dict(cur_data)                            # update_data, dfvo.py:264-287
        ref_data["flow"] = cur_data["flow"] = ref_data["flow_diff"] = None
    return np.stack(poses), modes


def test_mirrors_from_weight_files_reproduce_reference_main_loop(gpu, tmp_path):
    import os
    from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, write_weight_files
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dfvo_main.npz"))
    h, w, n = int(fx["h"]), int(fx["w"]), int(fx["n_frames"])
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=int(fx["seq_seed"]))
    assert np.array_equal(seq["poses"], fx["gt"])
    flow_path, depth_dir = write_weight_files(str(tmp_path), crafted_liteflownet_state_dict(h, w, "mux"),
                                              crafted_monodepth2_state_dict())
    cfg = full_cfg(h, w, flow_path, depth_dir)
    mirrors = _build_mirrors(cfg, seq["K"])
    deep_models = mirrors[0]
    assert deep_models.depth.feed_height == 192 and deep_models.depth.feed_width == 640
    assert tuple(deep_models.flow.get_target_size(h, w)) == (h, w)
    assert deep_models.conv_precision == "f16x3" and deep_models.session is not None  # what an unmodified apis/run.py gets
    poses, modes = _main_loop(cfg, seq, n, h, w, mirrors)
    st = deep_models.session.stats
    print("   modes", modes, "| final t (mirrors)", poses[-1][:3, 3], "(reference DFVO.main)", fx["poses"][-1][:3, 3], "| session", st)
    # the frame loop is exactly the call order the session is built for: every pair must have taken the resident paths
    assert st["push"] == n and st["flow_resident"] == n - 1 and st["kp_resident"] == n - 1 and st["pose_resident"] == n - 1
    assert st["pose_ahead"] == n - 1  # ... and every compute_pose_2d2d had been enqueued from forward_flow
    assert st["flow_plain"] == st["kp_plain"] == st["pose_plain"] == 0
    assert modes == list(fx["modes"])
    # the nets differ from the reference's torch-CPU nets in fp32 summation order (<= 2e-3 px on the flow), which can move
    # a keypoint across a threshold and with it the RANSAC samples: poses agree to the solver's noise level, not bit for bit
    for i in range(n):
        assert np.abs(poses[i][:3, :3] - fx["poses"][i][:3, :3]).max() < 1e-3
        assert np.linalg.norm(poses[i][:3, 3] - fx["poses"][i][:3, 3]) < 0.02 * max(1.0, np.linalg.norm(fx["poses"][i][:3, 3]))


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_frame_session_returns_what_the_plain_entry_points_return(gpu, tmp_path, monkeypatch, precision):
    """The session (libs/deep_models/session.py, csrc/session.hip) changes WHEN things run, never what a call returns: the
    same frame loop with DFVO_SESSION=0 (every call a blocking host-array entry point, as in rounds 1-4) and with the session
    gives bit-identical depth, flows, consistency map, keypoints, inlier masks, poses, scales and numpy RandomState at every
    pair.  Also: a caller that breaks the expected order or edits an array gets the plain path, not a stale result."""
    from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, write_weight_files
    h, w, n = 256, 640, 5
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=33)
    flow_path, depth_dir = write_weight_files(str(tmp_path), crafted_liteflownet_state_dict(h, w, "mux"),
                                              crafted_monodepth2_state_dict())
    cfg = full_cfg(h, w, flow_path, depth_dir)
    cfg["dfvo_hip"] = {"conv_precision": precision}
    runs = {}
    capi = importlib.import_module("df-vo_amd.capi")
    for sess in ("0", "1"):
        monkeypatch.setenv("DFVO_SESSION", sess)
        before = capi.lib().dfvo_get_conv_precision()
        mirrors = _build_mirrors(cfg, seq["K"])
        assert (mirrors[0].session is not None) == (sess == "1") and mirrors[0].conv_precision == precision
        # (initialize_models sets the process-wide packing precision for ITS nets and puts it back)
        assert capi.lib().dfvo_get_conv_precision() == before
        rec = []
        poses, modes = _main_loop(cfg, seq, n, h, w, mirrors, record=rec)
        runs[sess] = (poses, modes, rec, mirrors)
    (p0, m0, r0, _), (p1, m1, r1, mir) = runs["0"], runs["1"]
    assert m0 == m1 and np.array_equal(p0, p1)
    for a, b in zip(r0, r1):
        for k in a:
            assert np.array_equal(a[k], b[k]), "pair result '%s' differs between the plain entry points and the session" % k
    st = mir[0].session.stats
    assert st["flow_resident"] == st["kp_resident"] == st["pose_resident"] == st["pose_ahead"] == n - 1, st
    # ---- misuse: the session must fall back, never answer from stale state
    deep_models, sampler, e_tracker, _, _ = mir
    s = deep_models.session
    cur = {"id": 1, "img": seq["frames"][1].copy()}
    ref = {"id": 0, "img": seq["frames"][0].copy()}
    deep_models.forward_depth(imgs=[cur["img"]])            # pushed frame 1 after frame n - 1: (n - 1, 1) is not (0, 1)
    before = dict(s.stats)
    flows = deep_models.forward_flow(cur, ref, True)
    assert s.stats["flow_plain"] == before["flow_plain"] + 1
    assert np.array_equal(flows[(0, 1)], r0[0]["fwd"]) and np.array_equal(flows[(0, 1, "diff")], r0[0]["diff"])
    # an edited flow array loses its token: kp_selection runs on what it is given
    deep_models.forward_depth(imgs=[ref["img"]])
    deep_models.forward_depth(imgs=[cur["img"]])
    flows = deep_models.forward_flow(cur, ref, True)
    fwd, diff = flows[(0, 1)].copy(), flows[(0, 1, "diff")].copy()
    assert getattr(fwd, "_dfvo_tok", None) is not None
    diff[40:60, 100:200, 0] = 7.0                           # (a write through the array drops the token)
    assert getattr(diff, "_dfvo_tok", None) is None
    depth = np.ones((h, w))
    before = dict(s.stats)
    out = sampler.kp_selection({"depth": depth}, {"flow": fwd, "flow_diff": diff, "depth": depth})
    assert s.stats["kp_plain"] == before["kp_plain"] + 1
    want = T.local_bestN(np.asarray(fwd), np.asarray(diff))
    assert np.array_equal(out["kp1_best"], want["kp1_best"]) and np.array_equal(out["kp2_best"], want["kp2_best"])
    # np.random consumed between forward_flow (where the RandomState-consuming half of compute_pose_2d2d was enqueued ahead)
    # and compute_pose_2d2d: the result enqueued ahead is discarded, the call runs under the state it was made in
    deep_models.forward_depth(imgs=[ref["img"]])
    deep_models.forward_depth(imgs=[cur["img"]])
    flows = deep_models.forward_flow(cur, ref, True)
    sel = sampler.kp_selection({"depth": depth}, {"flow": flows[(0, 1)].copy(), "flow_diff": flows[(0, 1, "diff")].copy(), "depth": depth})
    np.random.seed(9)
    np.random.randint(0, 10, 3)
    st_call = np.random.get_state()
    before = dict(s.stats)
    got = e_tracker.compute_pose_2d2d(sel["kp1_best"][0], sel["kp2_best"][0], True)
    st_after = np.random.get_state()
    assert s.stats["pose_plain"] == before["pose_plain"] + 1 and s.stats["pose_ahead"] == before["pose_ahead"]
    np.random.set_state(st_call)
    res = T.compute_pose_2d2d(np.array(sel["kp1_best"][0]), np.array(sel["kp2_best"][0]), seq["K"])
    assert np.array_equal(got["pose"].R, res["R"]) and np.array_equal(got["pose"].t, res["t"])
    assert np.array_equal(st_after[1], np.random.get_state()[1]) and st_after[2] == np.random.get_state()[2]
    # ... and undisturbed, the same pair is answered from what was enqueued ahead, with the same bits as the plain path gives
    deep_models.forward_depth(imgs=[ref["img"]])
    deep_models.forward_depth(imgs=[cur["img"]])
    np.random.set_state(st_call)
    flows = deep_models.forward_flow(cur, ref, True)
    sel2 = sampler.kp_selection({"depth": depth}, {"flow": flows[(0, 1)].copy(), "flow_diff": flows[(0, 1, "diff")].copy(), "depth": depth})
    before = dict(s.stats)
    got2 = e_tracker.compute_pose_2d2d(sel2["kp1_best"][0], sel2["kp2_best"][0], True)
    assert s.stats["pose_ahead"] == before["pose_ahead"] + 1
    assert np.array_equal(got2["pose"].R, res["R"]) and np.array_equal(got2["pose"].t, res["t"]) and np.array_equal(got2["inliers"], got["inliers"])
    assert np.array_equal(st_after[1], np.random.get_state()[1]) and st_after[2] == np.random.get_state()[2]
    # keypoints other than the device's: compute_pose_2d2d takes the plain path and still matches the oracle
    np.random.seed(5)
    kp1, kp2 = out["kp1_best"][0][::2].copy(), out["kp2_best"][0][::2].copy()
    before = dict(s.stats)
    got = e_tracker.compute_pose_2d2d(kp1, kp2, True)
    assert s.stats["pose_plain"] == before["pose_plain"] + 1
    np.random.seed(5)
    res = T.compute_pose_2d2d(kp1, kp2, seq["K"])
    assert np.array_equal(got["pose"].R, res["R"]) and np.array_equal(got["pose"].t, res["t"])


def test_frame_session_over_a_long_sequence(gpu, tmp_path, monkeypatch):
    """The same comparison over 48 pairs (the host buffer rings wrap sixteen times; the early pose half and the per-keypoint depth
    entry of the scale recovery run on every pair -- this world never needs the PnP fallback): poses, tracking modes and the
    numpy RandomState after the last pair bit-identical with and without the session, every pair through the resident paths."""
    from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, write_weight_files
    h, w, n = 256, 640, 49
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=35)
    flow_path, depth_dir = write_weight_files(str(tmp_path), crafted_liteflownet_state_dict(h, w, "mux"),
                                              crafted_monodepth2_state_dict())
    cfg = full_cfg(h, w, flow_path, depth_dir)
    runs = {}
    for sess in ("0", "1"):
        monkeypatch.setenv("DFVO_SESSION", sess)
        mirrors = _build_mirrors(cfg, seq["K"])
        poses, modes = _main_loop(cfg, seq, n, h, w, mirrors)
        runs[sess] = (poses, modes, np.random.get_state(), mirrors[0].session)
    (p0, m0, st0, _), (p1, m1, st1, s) = runs["0"], runs["1"]
    print("   modes E / PnP:", m1.count("Ess. Mat."), m1.count("PnP"), "| session", s.stats, "| final t", p1[-1][:3, 3])
    assert m0 == m1 and np.array_equal(p0, p1)
    assert np.array_equal(st0[1], st1[1]) and st0[2] == st1[2]
    st = s.stats
    assert st["push"] == n and st["flow_resident"] == st["kp_resident"] == st["pose_resident"] == st["pose_ahead"] == n - 1, st
    assert st["flow_plain"] == st["kp_plain"] == st["pose_plain"] == 0
    assert np.linalg.norm(p1[-1][:3, 3]) > 10.0  # the sequence really moved


def test_frame_session_with_pnp_fallbacks(gpu, tmp_path, monkeypatch):
    """bench.py's class-surface workload (1241 x 376 'pot'-coded frames of a lateral motion, tracked ping-pong A -> B, B -> A: a few
    pairs in ten reject the E-tracker's pose or its scale and take the PnP fallback, through dfvo_compute_pose_3d2d_at_kp and with
    the early pose half discarded or consumed before it) -- session on and off: poses, modes and RandomState bit-identical"""
    from synth import (coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, tunnel_poses_lateral,
                       write_weight_files)
    h, w, n = 376, 1241, 31
    two = coded_tunnel_sequence(h, w, 2, mode="pot", step=1.0, seed=7, poses=tunnel_poses_lateral(2, 0.4))
    seq = {"frames": [two["frames"][i % 2] for i in range(n)], "K": two["K"]}
    flow_path, depth_dir = write_weight_files(str(tmp_path), crafted_liteflownet_state_dict(h, w, "pot"),
                                              crafted_monodepth2_state_dict())
    cfg = full_cfg(h, w, flow_path, depth_dir)
    runs = {}
    for sess in ("0", "1"):
        monkeypatch.setenv("DFVO_SESSION", sess)
        mirrors = _build_mirrors(cfg, seq["K"])
        poses, modes = _main_loop(cfg, seq, n, h, w, mirrors)
        runs[sess] = (poses, modes, np.random.get_state(), mirrors[0].session)
    (p0, m0, st0, _), (p1, m1, st1, s) = runs["0"], runs["1"]
    print("   modes E / PnP:", m1.count("Ess. Mat.") - 1, m1.count("PnP"), "| session", s.stats)
    assert m0 == m1 and np.array_equal(p0, p1)
    assert np.array_equal(st0[1], st1[1]) and st0[2] == st1[2]
    assert s.stats["pose_ahead"] == n - 1 and s.stats["pose_plain"] == 0
    assert m1.count("PnP") >= 1, "this workload is expected to exercise the PnP fallback"


def test_trajectory_composition_on_the_device(gpu):
    """update_global_pose over a gathered sequence in one launch (dfvo_compose_trajectory) against the host
    loop of dist.compose_trajectory -- E / PnP rows, constant-motion rows (status 1: the previous motion is reused, also
    when several follow each other and at row 0), a non-identity first pose, host arrays and a CUDA tensor; a status-2 row
    is refused with the row index"""
    import torch
    dmod = importlib.import_module("df-vo_amd.dist")
    rng = np.random.Generator(np.random.PCG64(11))
    n = 301
    rows = np.zeros((n, 17))
    for i in range(n):
        w = rng.normal(0, 0.01, 3)
        th = np.linalg.norm(w)
        Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, rng.normal(0, 0.5, 3) + [0, 0, 1.0]
        rows[i, :16] = T.reshape(-1)
    rows[[0, 7, 8, 9, 100], 16] = 1       # constant motion (row 0: the identity "previous" motion)
    rows[[20, 21], 16] = 3                # PnP rows compose like E rows
    first = np.eye(4)
    first[:3, 3] = [1.0, -2.0, 3.0]
    for fp in (None, first):
        want = dmod.compose_trajectory(rows, fp)
        got_h = dmod.compose_trajectory_device(rows, fp)
        got_d = dmod.compose_trajectory_device(torch.from_numpy(rows).cuda(), fp)
        assert got_h.shape == want.shape == (n + 1, 4, 4)
        assert np.array_equal(got_h, got_d)
        assert np.abs(got_h - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    rows[50, 16] = 2
    with pytest.raises(ValueError, match="pair 50"):
        dmod.compose_trajectory_device(rows)


class _RefTimer:
    """the surface of the reference's libs/general/timer.py Timer that the trackers touch (timers dict, add / start / end)"""

    def __init__(self):
        self.timers = {}

    def add(self, item, group=None):
        self.timers[item] = {"name": item, "time": 0, "is_counting": False, "duration": [], "group": group}

    def start(self, item, group=None):
        if self.timers.get(item, -1) == -1:
            self.add(item, group)
        assert not self.timers[item]["is_counting"]
        self.timers[item]["is_counting"] = True

    def end(self, item):
        assert self.timers[item]["is_counting"]
        self.timers[item]["is_counting"] = False


def test_timer_sub_keys_are_fed_from_device_stage_times(gpu):
    """`timers` is part of EssTracker's signature.  The reference times GRIC-H / find H / find-Ess / GRIC-E /
    find-Ess (full) / recover pose / triangulation / scale ransac around Python statements (E_tracker.py:197-296,597-638);
    here the stages are kernels behind one C call and the mirror appends their HIP-event durations under the same keys and
    groups (dfvo_tracker_stage_ms)."""
    h, w = 192, 640
    cfg = make_cfg(h, w)
    fr = rigid_scene(h, w, seed=300, bad_frac=0.3)
    K = fr["K"]
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    ks_mod = importlib.import_module("df-vo_amd.libs.matching.keypoint_sampler")
    trk_mod = importlib.import_module("df-vo_amd.libs.tracker")
    cam = cam_mod.Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
    ref_timer = "/root/reference/libs/general/timer.py"
    import os
    if os.path.exists(ref_timer):  # build container with a GPU: the reference's own class
        from importlib import util as ilu
        spec = ilu.spec_from_file_location("ref_timer", ref_timer)
        mod = ilu.module_from_spec(spec)
        spec.loader.exec_module(mod)
        timers = mod.Timer()
    else:
        timers = _RefTimer()
    np.random.seed(4869)
    trk = trk_mod.EssTracker(cfg, cam, timers)
    frames = [fr, rigid_scene(h, w, seed=301, bad_frac=0.4)]
    got = track_sequence(frames, cfg, ks_mod.KeypointSampler(cfg), trk, trk_mod.PnpTracker(cfg, cam), cam_mod.SE3)
    assert all(m in ("Ess. Mat.", "PnP") for m, _ in got)  # either way the E-tracker ran twice
    # the scale recovery on its own (dfvo.py:198 runs it only behind a non-zero E translation): twice, on frame 0's keypoints
    sampler = ks_mod.KeypointSampler(cfg)
    ref = {"flow": fr["flow"], "flow_diff": fr["diff"][..., None], "depth": fr["depth_ref"], "raw_depth": fr["depth_ref"].astype(np.float32)}
    cur = {"depth": fr["depth_cur"]}
    sampler.update_kp_data(cur, ref, sampler.kp_selection(cur, ref))
    pose = cam_mod.SE3()
    pose.t = np.array([[0.02], [0.01], [0.8]])
    for _ in range(2):
        trk.scale_recovery(cur, ref, pose, False)
    for key in ("triangulation", "scale ransac"):  # (the tracked pairs may have added entries of their own: keep the last two)
        timers.timers[key]["duration"] = timers.timers[key]["duration"][-2:]
    groups = {"find H": "E-tracker", "GRIC-H": "E-tracker", "find-Ess": "E-tracker", "GRIC-E": "E-tracker",
              "find-Ess (full)": "E-tracker", "recover pose": "E-tracker", "triangulation": "scale_recovery",
              "scale ransac": "scale_recovery"}
    for key, grp in groups.items():
        t = timers.timers[key]
        assert t["group"] == grp and len(t["duration"]) == 2 and not t["is_counting"], key
        assert all(1e-6 < d < 0.5 for d in t["duration"]), (key, t["duration"])
    d = {k: timers.timers[k]["duration"][1] for k in groups}
    assert d["GRIC-H"] >= d["find H"] and d["find-Ess (full)"] >= d["find-Ess"] + d["GRIC-E"] * 0.5
    print("stage ms:", {k: round(v * 1e3, 3) for k, v in d.items()})
    # a pair the E-tracker never reaches (fewer than 11 keypoints): the E keys get no new entry
    small = np.ascontiguousarray(np.random.rand(8, 2) * 100)
    trk.compute_pose_2d2d(small, small + 1.0, True)
    assert len(timers.timers["find-Ess"]["duration"]) == 2 and len(timers.timers["find H"]["duration"]) == 3


# ---- round 6: guards of the frame session (VERDICT r5 tasks 5 / 8) ----------------------------------------------
def _small_world(tmp_path, n=6, seed=41, tweak=None, precision=None):
    from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, write_weight_files
    h, w = 256, 640
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=seed)
    fsd, dsd = crafted_liteflownet_state_dict(h, w, "mux"), crafted_monodepth2_state_dict()
    if tweak:
        tweak(fsd, dsd)
    flow_path, depth_dir = write_weight_files(str(tmp_path), fsd, dsd)
    cfg = full_cfg(h, w, flow_path, depth_dir)
    if precision:
        cfg["dfvo_hip"] = {"conv_precision": precision}
    return h, w, seq, cfg


def test_f16_range_guard_raises_through_the_mirrors(gpu, tmp_path, monkeypatch):
    """DeepModel packs the nets in f16x3 by default; an activation beyond +-65504 becomes inf in the layer that splits it.  The
    mirrors must not hand such a map out: forward_depth / forward_flow raise DfvoError, with the session (counter read behind
    each net) and without it (counter read after the call); the exact-fp32 packing of the same weights runs through."""
    capi = importlib.import_module("df-vo_amd.capi")

    def hot_depth(fsd, dsd):
        dsd["encoder.bn1.bias"][:] = 1.0e5      # conv1 + BN + ReLU outputs ~1e5: the first residual block splits them

    def hot_flow(fsd, dsd):
        fsd["moduleFeatures.moduleOne.0.bias"][:] = 1.0e5

    for sess in ("1", "0"):
        monkeypatch.setenv("DFVO_SESSION", sess)
        h, w, seq, cfg = _small_world(tmp_path, tweak=hot_depth)
        dm = _build_mirrors(cfg, seq["K"])[0]
        assert (dm.session is not None) == (sess == "1") and dm.conv_precision == "f16x3"
        with pytest.raises(capi.DfvoError, match="out of range"):
            dm.forward_depth(imgs=[seq["frames"][0].copy()])
        h, w, seq, cfg = _small_world(tmp_path, tweak=hot_flow)
        dm = _build_mirrors(cfg, seq["K"])[0]
        f0, f1 = seq["frames"][0].copy(), seq["frames"][1].copy()
        dm.forward_depth(imgs=[f0])             # (the depth net of these weights is in range)
        with pytest.raises(capi.DfvoError, match="out of range"):
            # with the session the flow net of (f0, f1) runs beside the depth net of f1 and the counter is process-wide: the
            # event may already fail forward_depth(f1); without it, forward_flow is the call that runs the flow net
            dm.forward_depth(imgs=[f1])
            dm.forward_flow({"id": 1, "img": f1}, {"id": 0, "img": f0}, True)
    # the same hot weights in exact fp32: no split, no guard, finite output
    monkeypatch.setenv("DFVO_SESSION", "1")
    h, w, seq, cfg = _small_world(tmp_path, tweak=hot_depth, precision="fp32")
    dm = _build_mirrors(cfg, seq["K"])[0]
    assert np.isfinite(np.asarray(dm.forward_depth(imgs=[seq["frames"][0].copy()]))).all()
    capi.f16s_overflow_count(reset=True)        # (the suite asserts a zero counter after every test)


def test_session_compares_whole_frames_and_whole_flow_arrays(gpu, tmp_path, monkeypatch):
    """Round 5 validated the hand-over on a 1024-element sample: an edit elsewhere went unnoticed and the stale flow / the
    keypoints of the unedited flow came back.  Now: (a) one byte of the current frame changed in place between forward_depth
    and forward_flow -> forward_flow takes the plain path and returns the flow of the frames it was GIVEN; (b) a copy of the
    returned flow edited through a child view (no __setitem__ on the array object, the token survives) -> kp_selection takes
    the plain path and selects on what it was given."""
    monkeypatch.setenv("DFVO_SESSION", "1")
    h, w, seq, cfg = _small_world(tmp_path)
    dm, sampler, _, _, _ = _build_mirrors(cfg, seq["K"])
    s = dm.session
    f0, f1 = seq["frames"][0].copy(), seq["frames"][1].copy()
    dm.forward_depth(imgs=[f0])
    dm.forward_depth(imgs=[f1])
    f1.reshape(-1)[5] ^= 0x40                    # flat index 5: in no strided sample of round 5's check
    before = dict(s.stats)
    flows = dm.forward_flow({"id": 1, "img": f1}, {"id": 0, "img": f0}, True)
    assert s.stats["flow_plain"] == before["flow_plain"] + 1 and s.stats["flow_resident"] == before["flow_resident"]
    want = dm.flow.inference_flow_u8(f0, f1)
    assert np.array_equal(flows[(0, 1)], want[0]) and np.array_equal(flows[(0, 1, "diff")], want[2])
    # (b)
    g0, g1 = seq["frames"][2].copy(), seq["frames"][3].copy()
    dm.forward_depth(imgs=[g0])
    dm.forward_depth(imgs=[g1])
    flows = dm.forward_flow({"id": 3, "img": g1}, {"id": 2, "img": g0}, True)
    assert s.stats["flow_resident"] == before["flow_resident"] + 1
    fwd, diff = flows[(2, 3)].copy(), flows[(2, 3, "diff")].copy()
    child = diff[100:140]                        # a view: writing through it leaves the parent's token alone
    np.asarray(child)[...] = 9.0
    assert getattr(diff, "_dfvo_tok", None) is not None, "the token is expected to survive a write through a child view"
    depth = np.ones((h, w))
    before = dict(s.stats)
    out = sampler.kp_selection({"depth": depth}, {"flow": fwd, "flow_diff": diff, "depth": depth})
    assert s.stats["kp_plain"] == before["kp_plain"] + 1 and s.stats["kp_resident"] == before["kp_resident"]
    want = T.local_bestN(np.asarray(fwd), np.asarray(diff))
    assert np.array_equal(out["kp1_best"], want["kp1_best"]) and np.array_equal(out["kp2_best"], want["kp2_best"])


def test_session_views_keep_their_memory(gpu, tmp_path, monkeypatch):
    """forward_depth / forward_flow return views of a three-slot pinned ring.  A view that is still referenced when
    its slot comes up for reuse keeps the memory (the session takes fresh buffers); after close() it still reads what it read
    before.  The arrays are read-only."""
    monkeypatch.setenv("DFVO_SESSION", "1")
    h, w, seq, cfg = _small_world(tmp_path, n=8)
    dm = _build_mirrors(cfg, seq["K"])[0]
    s = dm.session
    frames = [f.copy() for f in seq["frames"]]
    d0 = dm.forward_depth(imgs=[frames[0]])
    d1 = dm.forward_depth(imgs=[frames[1]])
    fl = dm.forward_flow({"id": 1, "img": frames[1]}, {"id": 0, "img": frames[0]}, True)
    held = {"d0": d0, "d1": d1[10:20], "fwd": fl[(0, 1)][0], "diff": fl[(0, 1, "diff")]}
    snap = {k: np.array(v) for k, v in held.items()}
    with pytest.raises(ValueError):
        d0[0, 0] = 1.0
    del d0, d1, fl
    for k in range(2, 8):                        # six more pushes: both slots are reused twice
        dm.forward_depth(imgs=[frames[k]])
        dm.forward_flow({"id": k, "img": frames[k]}, {"id": k - 1, "img": frames[k - 1]}, True)
    assert s.stats["slots_detached"] == 2, s.stats
    for k, v in held.items():
        assert np.array_equal(np.asarray(v), snap[k]), "'%s' changed under its holder" % k
    last = dm.forward_depth(imgs=[frames[0]])
    last_snap = np.array(last)
    s.close()
    assert np.array_equal(np.asarray(last), last_snap)
    for k, v in held.items():
        assert np.array_equal(np.asarray(v), snap[k])
    # a closed session no longer accepts frames: the DeepModel answers from the plain entry point
    again = dm.forward_depth(imgs=[frames[0]])
    assert np.array_equal(np.asarray(again), last_snap)


def test_depth_only_caller_stops_paying_for_the_flow_net(gpu, tmp_path, monkeypatch):
    """every push enqueued the flow net whether or not forward_flow followed.  After three pushes in a row whose
    flow nobody asked for, a push runs the depth net alone; the first forward_flow is answered by the plain entry point (same
    bits) and switches the speculation back on."""
    monkeypatch.setenv("DFVO_SESSION", "1")
    h, w, seq, cfg = _small_world(tmp_path, n=8)
    dm = _build_mirrors(cfg, seq["K"])[0]
    s = dm.session
    frames = [f.copy() for f in seq["frames"]]
    for k in range(6):
        dm.forward_depth(imgs=[frames[k]])
    assert s.stats["push"] == 6 and s.stats["push_no_flow"] == 3, s.stats
    before = dict(s.stats)
    flows = dm.forward_flow({"id": 5, "img": frames[5]}, {"id": 4, "img": frames[4]}, True)
    assert s.stats["flow_plain"] == before["flow_plain"] + 1
    want = dm.flow.inference_flow_u8(frames[4], frames[5])
    assert np.array_equal(flows[(4, 5)], want[0])
    dm.forward_depth(imgs=[frames[6]])           # speculating again; (5, 6) runs both frames through Features
    flows = dm.forward_flow({"id": 6, "img": frames[6]}, {"id": 5, "img": frames[5]}, True)
    assert s.stats["flow_resident"] == before["flow_resident"] + 1 and s.stats["push_no_flow"] == 3
    want = dm.flow.inference_flow_u8(frames[5], frames[6])
    assert np.array_equal(flows[(5, 6)], want[0]) and np.array_equal(flows[(5, 6, "diff")], want[2])


@pytest.mark.parametrize("mode", ["force_fail", "creation"])
def test_stream_pool_fallback_is_loud_and_harmless(gpu, tmp_path, monkeypatch, capfd, mode):
    """VERDICT r5 task 8: the pipe probe is a timing measurement.  Forced to fail (DFVO_STREAM_POOL_FORCE_FAIL=1) the session
    keeps creation-order streams and says so on stderr, once; DFVO_STREAM_POOL=creation skips the probe silently.  Either way
    every call returns what the plain entry points return."""
    if mode == "force_fail":
        monkeypatch.setenv("DFVO_STREAM_POOL_FORCE_FAIL", "1")
    else:
        monkeypatch.setenv("DFVO_STREAM_POOL", "creation")
    h, w, seq, cfg = _small_world(tmp_path, n=4)
    runs = {}
    for sess in ("0", "1"):
        monkeypatch.setenv("DFVO_SESSION", sess)
        mirrors = _build_mirrors(cfg, seq["K"])
        rec = []
        poses, modes = _main_loop(cfg, seq, 4, h, w, mirrors, record=rec)
        runs[sess] = (poses, modes, rec, mirrors[0].session)
    err = capfd.readouterr().err
    assert ("pipe probe did not settle" in err) == (mode == "force_fail"), err[-400:]
    (p0, m0, r0, _), (p1, m1, r1, s) = runs["0"], runs["1"]
    assert m0 == m1 and np.array_equal(p0, p1)
    for a, b in zip(r0, r1):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert s.stats["flow_resident"] == s.stats["pose_ahead"] == 3, s.stats
