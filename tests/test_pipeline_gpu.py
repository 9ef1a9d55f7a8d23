"""GPU: the fused per-pair pipeline (dfvo_pipeline_*) against (a) the stand-alone net executors it is
built from and (b) the oracle's tracker chain on a synthetic rigid scene (identical flow / consistency /
depth maps and identical numpy RandomState)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from oracle import nets_torch as O
from oracle import tracker_np as T
from synth import image_pair, rigid_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe_mod(gpu):
    return importlib.import_module("df-vo_amd.pipeline")


@pytest.mark.parametrize("h,w", [(192, 640), (376, 1241)])
def test_pipeline_tracker_matches_oracle_chain(gpu, conv_precision, pipe_mod, h, w):
    sc = rigid_scene(h, w, seed=3 + h)
    K = sc["K"]
    pipe = pipe_mod.TrackingPipeline(h, w, 192, 640, K, O.liteflownet_state_dict(4869), O.monodepth2_state_dict(4869),
                                     seed=4869)
    ref, cur = image_pair(h, w, seed=11)
    feed, _ = image_pair(192, 640, seed=12)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dref, dcur, dfeed = d(ref), d(cur), d(feed)
    dflow, ddiff, ddepth = d(sc["flow"]), d(sc["diff"]), d(sc["depth_cur"])
    poses = []
    np.random.seed(4869)
    pipe.set_ref_depth(depth=d(sc["depth_ref"]))  # depth of the first reference frame (PnP fallback input)
    ref_depth = sc["depth_ref"]
    for frame in range(3):  # the RandomState carries over from pair to pair, as in a sequence
        pipe.enqueue_nets(frame % 2, dref, dcur, dfeed)
        if frame == 1:  # one pair through the prefetch path (keypoints + homography chain enqueued ahead of track)
            pipe.prefetch_track(frame % 2, dflow, ddiff)
        out = pipe.track(frame % 2, dflow, ddiff, ddepth)
        kp = T.local_bestN(sc["flow"], sc["diff"][..., None])
        assert out.good_kp_found == int(kp["good_kp_found"]) and out.n_kp == kp["kp1_best"].shape[1]
        res = T.compute_pose_2d2d(kp["kp1_best"][0], kp["kp2_best"][0], K)
        R = np.array(out.R[:]).reshape(3, 3)
        t = np.array(out.t[:]).reshape(3, 1)
        if out.status == 0:  # (with status 3 the E-tracker pose has been replaced by the PnP one, checked below)
            assert np.array_equal(R, res["R"]) and np.array_equal(t, res["t"]), "frame %d" % frame
        pose = np.eye(4)
        pose[:3, :3] = res["R"]
        pose[:3, 3:] = res["t"]
        diag = {"n_valid": -1}
        s_ref = -1
        if np.linalg.norm(res["t"]) != 0:  # dfvo.py:198: scale recovery (and its RandomState draws) only then
            s_ref = T.find_scale_from_depth(kp["kp1_best"][0], kp["kp2_best"][0], np.linalg.inv(pose), sc["depth_cur"], K,
                                            diag=diag)
        print("frame %d: status %d kp %d inliers %d | scale hip %.12g oracle %.12g (valid %d/%d trials %d/%d)" % (
            frame, out.status, out.n_kp, out.best_inlier_cnt, out.scale, s_ref, out.scale_n_valid, diag["n_valid"],
            out.scale_n_trials, diag.get("n_trials", 0)))
        if np.linalg.norm(res["t"]) == 0 or s_ref == -1:  # E rejected (GRIC / cheirality) or no scale: PnP fallback
            assert out.status == 3
            pnp = T.compute_pose_3d2d(kp["kp1_best"][0], kp["kp2_best"][0], ref_depth, K, 0.0, 50.0, 5, 100, 1.0)
            assert out.pnp_n_filtered == len(pnp["kp1"]) and out.pnp_inliers == pnp["best_inlier"]
            assert np.array_equal(np.array(out.R[:]).reshape(3, 3), pnp["R"])
            assert np.array_equal(np.array(out.t[:]).reshape(3, 1), pnp["t"])
            rel, mode = pipe.hybrid_pose(out, np.eye(4))
            assert mode == "PnP" and np.abs(rel - pnp["pose"]).max() <= 1e-12
            ref_depth = sc["depth_cur"]
            continue
        ref_depth = sc["depth_cur"]  # the current depth rolls over to the reference slot
        assert out.status == 0
        assert out.scale_n_valid == diag["n_valid"]
        assert abs(out.scale - s_ref) <= 1e-9 * abs(s_ref)
        # geometry sanity: recovered motion is the inverse of the simulated ref -> cur motion
        Tgt = np.eye(4)
        Tgt[:3, :3] = sc["R"]
        Tgt[:3, 3] = sc["t"]
        Tinv = np.linalg.inv(Tgt)
        rel, _ = pipe.hybrid_pose(out, np.eye(4))
        assert np.abs(rel[:3, :3] - Tinv[:3, :3]).max() < 5e-3
        assert np.linalg.norm(rel[:3, 3] - Tinv[:3, 3]) < 0.15 * np.linalg.norm(Tinv[:3, 3])
        poses.append(rel)
    pipe.close()


def test_pipeline_nets_equal_standalone_executors(gpu, conv_precision, pipe_mod):
    h, w = 192, 640
    lib = gpu.lib()
    fsd, dsd = O.liteflownet_state_dict(4869), O.monodepth2_state_dict(4869)
    K = rigid_scene(64, 64, seed=1)["K"]
    pipe = pipe_mod.TrackingPipeline(h, w, 192, 640, K, fsd, dsd)
    ref, cur = image_pair(h, w, seed=21)
    feed, _ = image_pair(192, 640, seed=22)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pipe.enqueue_nets(1, d(ref), d(cur), d(feed))
    pipe.sync()
    fwd, bwd, diff, raw, dep = pipe.get_outputs(1)
    pipe.close()
    # stand-alone executors through the mirror classes
    lf = importlib.import_module("df-vo_amd.libs.deep_models.flow.lite_flow_net.lite_flow").LiteFlow(h, w)
    lf.initialize_network_model(fsd, False)
    f2, b2, d2 = lf.inference_flow_u8(ref, cur)
    assert np.array_equal(fwd, f2) and np.array_equal(bwd, b2) and np.array_equal(diff, d2[..., 0])
    md = importlib.import_module("df-vo_amd.libs.deep_models.depth.monodepth2.monodepth2").Monodepth2DepthNet(h, w)
    enc = {k: v for k, v in dsd.items() if k.startswith("encoder.")}
    enc.update(height=192, width=640)
    md.initialize_network_model({"encoder": enc, "decoder": {k: v for k, v in dsd.items() if k.startswith("decoder.")}},
                                "kitti_odom", False)
    depth_small = md.inference_depth_u8(feed)
    from oracle import cv2_shim
    want_raw = cv2_shim.resize(depth_small, (w, h), interpolation=cv2_shim.INTER_NEAREST)
    assert np.array_equal(raw, want_raw)
    want = T.preprocess_depth(want_raw, [[0.3, 1], [0, 1]], [0, 50])
    assert np.array_equal(dep, want)
    # the tensor surface of the depth mirror (monodepth2.py:91-139, deep_depth.py:74-85): inference_depth = the device depth with
    # the stereo-baseline multiplier; inference / inference_no_grad = {'depth', 'disp'} of scale 0 without it; pred_* kept
    timg = torch.from_numpy(feed).permute(2, 0, 1)[None].float() / 255
    dd = md.inference_depth(timg)
    assert tuple(dd.shape) == (1, 1, 192, 640) and np.array_equal(dd[0, 0].numpy(), depth_small)
    outs = md.inference_no_grad(timg)
    assert set(outs) == {"depth", "disp"} and set(outs["depth"]) == {0} and outs["depth"][0].shape == dd.shape
    assert torch.equal(outs["depth"][0], dd / md.stereo_baseline_multiplier) and torch.equal(outs["disp"][0], 1.0 / outs["depth"][0])
    assert torch.equal(md.pred_depths[0], outs["depth"][0]) and torch.equal(md.pred_disps[0], outs["disp"][0])
    o_depth = O.depth_inference(dsd, feed)  # (the oracle's depth is the multiplied one too)
    assert np.abs(outs["depth"][0][0, 0].numpy() * 5.4 - o_depth).max() <= 1e-3 * np.abs(o_depth).max()
    # DeepFlow helpers of the flow mirror (deep_flow.py:107-129,171-196) on the device, against the torch restatements
    g = torch.Generator().manual_seed(3)
    fl1, fl2 = torch.randn(2, 2, 24, 40, generator=g) * 3, torch.randn(2, 2, 24, 40, generator=g) * 3
    got, want_t = lf.resize_dense_flow(fl1, 37, 53), O.resize_dense_flow(fl1, 37, 53)
    assert got.shape == want_t.shape and (got - want_t).abs().max() <= 1e-5 * want_t.abs().max()
    px = O.flow_to_pix(fl1)
    got, want_t = lf.forward_backward_consistency(fl1, fl2, px), O.forward_backward_consistency(fl1, fl2, px)
    assert got.shape == want_t.shape == (2, 24, 40, 1) and (got - want_t).abs().max() <= 1e-4 * max(1.0, float(want_t.abs().max()))
    with pytest.raises(NotImplementedError):
        lf.load_flow_file("x.npy")
    ks = importlib.import_module("df-vo_amd.libs.matching.keypoint_sampler").KeypointSampler
    assert ks.get_feat_track_methods(None, 1) == "deep_flow"


@pytest.mark.parametrize("h,w,instances", [(192, 640, "2"), (376, 1241, "2"), (192, 640, "1")])
def test_feature_carry_equals_recomputed_features(gpu, conv_precision, pipe_mod, h, w, instances, monkeypatch):
    """enqueue_nets(ref=None): the reference frame's image / feature pyramids come from the previous pass (the other flow-net
    instance with two instances, the same one with one) instead of a second Features run: forward / backward flow and the
    consistency map of the pair must be what the full pass on the same two frames gives, bit for bit (Features always runs
    one frame per launch, and the instances share their layer configurations: nets.hip enqueue_features_both, autotune_conv)."""
    monkeypatch.setenv("DFVO_FLOW_INSTANCES", instances)
    fsd, dsd = O.liteflownet_state_dict(4869), O.monodepth2_state_dict(4869)
    K = rigid_scene(64, 64, seed=1)["K"]
    pipe = pipe_mod.TrackingPipeline(h, w, 192, 640, K, fsd, dsd)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    a, b = image_pair(h, w, seed=31)
    c, e = image_pair(h, w, seed=32)
    frames = [d(x) for x in (a, b, c, e)]
    carried, full = [], []
    pipe.enqueue_nets(0, frames[0], frames[1])  # a sequence: first pair full, then carried, slot (= instance) alternating
    for j in (1, 2):
        pipe.enqueue_nets(j, None, frames[j + 1])
    pipe.sync()
    for j in (1, 2):
        carried.append(pipe.get_outputs(j)[:3])
    for j in (1, 2):  # the same pairs with both frames handed over
        pipe.enqueue_nets(j, frames[j], frames[j + 1])
        pipe.sync()
        full.append(pipe.get_outputs(j)[:3])
    # and carried again after full passes (the carry source is whichever instance ran last)
    pipe.enqueue_nets(3, None, frames[0])
    pipe.sync()
    again = pipe.get_outputs(3)[:3]
    pipe.enqueue_nets(0, frames[3], frames[0])
    pipe.sync()
    again_full = pipe.get_outputs(0)[:3]
    # a full pass enqueued right behind a carried one, no sync in between: it overwrites the pyramids the carried pass
    # (other instance) is still copying from unless the pipeline orders it behind that pass's feature stage
    for rep in range(3):
        pipe.enqueue_nets(0, frames[0], frames[1])
        pipe.enqueue_nets(1, None, frames[2])
        pipe.enqueue_nets(2, frames[2], frames[3])
        pipe.sync()
        carried.append(pipe.get_outputs(1)[:3])
        full.append(full[0])
        carried.append(pipe.get_outputs(2)[:3])
        full.append(full[1])
    pipe.close()
    worst = 0.0
    for got, want in zip(carried + [again], full + [again_full]):
        for g, wnt in zip(got, want):
            worst = max(worst, float(np.abs(g - wnt).max()))
    print("feature carry vs recomputed: max |diff| %.3g px (0 = bit-identical)" % worst)
    assert worst == 0.0


def test_track_begin_refuses_a_second_pending_pair(gpu, pipe_mod):
    """track_begin(k + 1) before track_end(k) would put pair k + 1's RandomState draws ahead of pair k's PnP decision: refused"""
    h, w = 192, 640
    sc = rigid_scene(h, w, seed=5)
    pipe = pipe_mod.TrackingPipeline(h, w, 192, 640, sc["K"], O.liteflownet_state_dict(4869), O.monodepth2_state_dict(4869), seed=1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ref, cur = image_pair(h, w, seed=11)
    dref, dcur = d(ref), d(cur)
    dflow, ddiff, ddepth = d(sc["flow"]), d(sc["diff"]), d(sc["depth_cur"])
    pipe.set_ref_depth(depth=d(sc["depth_ref"]))
    pipe.enqueue_nets(0, dref, dcur)
    pipe.enqueue_nets(1, dcur, dref)
    pipe.track_begin(0, dflow, ddiff, ddepth)
    with pytest.raises(gpu.DfvoError, match="not yet collected"):
        pipe.track_begin(1, dflow, ddiff, ddepth)
    out0 = pipe.track_end(0)
    pipe.track_begin(1, dflow, ddiff, ddepth)
    out1 = pipe.track_end(1)
    assert out0.status in (0, 3) and out1.status in (0, 3)
    pipe.close()
