"""GPU: the product data path with NO overrides -- uint8 frames -> HIP nets -> the nets' own flow / consistency /
depth -> keypoint selection -> E / PnP RANSAC -> pose (libs/dfvo.py:299-345 then :121-262).

Frames come from the coded tunnel world (df-vo_amd/synthetic.py): they carry their flow and depth fields as colour
codes and a few channels of otherwise random network weights decode them, so the real layer stacks emit coherent
rigid-scene flow and depth and `good_kp_found` is true.  Checks per pair, over consecutive pairs with the depth
roll-over and the numpy RandomState carried along:
  (1) HIP nets vs the torch-CPU oracle on the same frames (tolerance, absolute pixels / relative depth),
  (2) the solver stage vs the oracle chain run on the HIP nets' OWN output arrays: keypoints (values and order), inlier
      mask, R, t, scale, PnP result and the RandomState bit for bit,
  (3) the recovered motion vs the ground truth of the rendered sequence."""
import importlib

import numpy as np
import pytest
import torch

from oracle import nets_torch as O
from oracle import pipeline_np as P
from synth import (coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict, tunnel_poses_lateral,
                   tunnel_truth)

pytestmark = pytest.mark.gpu

FLOW_TOL_PX = 2e-3    # |HIP - oracle| on the 376x1241 / 192x640 flow fields, absolute pixels (fp32 summation order)
DEPTH_TOL_REL = 1e-3


_oracle_cache = {}
_LARGE = None


def _large_fixture():
    global _LARGE
    if _LARGE is None:
        import os
        _LARGE = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coded_large.npz"))
    return _LARGE


def _crc(a):
    import zlib
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff


def _oracle_pair(fsd, seq, h, w, mode, step, k):
    """(fwd, bwd, diff, pixel step): the torch-CPU oracle live, or -- BASELINE config 4 / 5 sizes -- the fixture it wrote in
    the build container (tests/golden/make_oracle_fixtures.py coded_large: every 8th pixel, frames pinned by CRC)"""
    if h * w > 384 * 1248:
        fx = _large_fixture()
        assert _crc(seq["frames"]) == int(fx["crc_%dx%d" % (h, w)]), "coded frames differ from the ones the fixture was computed on"
        key = "%dx%d_%d" % (h, w, k)
        return fx["fwd_" + key], fx["bwd_" + key], fx["diff_" + key][..., None], int(fx["step"])
    okey = (h, w, mode, step, k)
    if okey not in _oracle_cache:
        _oracle_cache[okey] = O.flow_inference(fsd, seq["frames"][k], seq["frames"][k + 1])
    return _oracle_cache[okey] + (1,)


def _mods():
    return importlib.import_module("df-vo_amd.pipeline"), importlib.import_module("df-vo_amd.sequence")


@pytest.mark.parametrize("h,w,mode,step", [(256, 640, "mux", 1.0), (192, 640, "pot", 0.3), (376, 1241, "pot", 0.3), (376, 1241, "pot", "lateral"),
                                              (384, 1248, "mux", 1.0), (960, 1280, "mux", 1.0), (1280, 1920, "mux", 1.0)])
def test_images_to_pose_no_overrides(gpu, conv_precision, h, w, mode, step):
    pmod, smod = _mods()
    large = h * w > 384 * 1248  # BASELINE configs 4 / 5 (RobotCar 1280x960, synthetic 1920x1280): oracle nets from the fixture
    n_frames = 3 if (step == "lateral" or large) else 4
    lateral = step == "lateral"  # sideways drive: the potential encoding then gives E-tracked pairs at KITTI size
    seq = (coded_tunnel_sequence(h, w, n_frames, mode=mode, poses=tunnel_poses_lateral(n_frames, 0.4)) if lateral else
           coded_tunnel_sequence(h, w, n_frames, mode=mode, step=step))
    fsd, dsd = crafted_liteflownet_state_dict(h, w, mode), crafted_monodepth2_state_dict()
    K = seq["K"]
    pipe = pmod.TrackingPipeline(h, w, 192, 640, K, fsd, dsd, seed=4869)
    fr = smod.frames_to_device(seq["frames"])
    # depth of frame 0 exactly as the device computes it (input of the first pair's PnP fallback)
    pipe.enqueue_nets(3, fr[0], fr[0])
    pipe.sync()
    depth_ref = pipe.get_outputs(3)[4]
    pipe.set_ref_image(fr[0])
    np.random.seed(4869)
    modes = []
    # from-images accounting (north_star: "bit-exact RANSAC inlier masks under a fixed seed, pose matrices within 1e-4
    # Frobenius on identical image pairs"): the oracle chain run on the ORACLE nets' outputs, its own RandomState carried
    # from pair to pair -- i.e. an independent run of the whole path from the same uint8 frames
    st_img = np.random.get_state()
    odepth_ref = None if large else P.frame_depth(dsd, seq["frames"][0])[1]
    acct = []
    for k in range(n_frames - 1):
        slot = k % 3
        pipe.enqueue_nets(slot, fr[k], fr[k + 1])
        if k == 1:
            pipe.prefetch_track(slot)  # one pair through the run-ahead path
        out = pipe.track(slot)
        fwd, bwd, diff, raw, dep = pipe.get_outputs(slot)
        kp_ref, kp_cur, inl = pipe.get_keypoints(slot)
        # (1) nets vs oracle nets
        ofwd, obwd, odiff, st = _oracle_pair(fsd, seq, h, w, mode, step, k)
        e_flow = max(np.abs(fwd[:, ::st, ::st] - ofwd).max(), np.abs(bwd[:, ::st, ::st] - obwd).max())
        assert e_flow <= FLOW_TOL_PX, "pair %d: |flow HIP - oracle| = %.3e px (max |flow| %.1f px)" % (k, e_flow, np.abs(ofwd).max())
        oraw, odep = P.frame_depth(dsd, seq["frames"][k + 1])
        assert np.abs(raw - oraw).max() <= DEPTH_TOL_REL * np.abs(oraw).max()
        gt_f, gt_b, z1 = tunnel_truth(seq, k)
        med = float(np.median(np.abs(fwd - gt_f)))
        assert med < 0.05, "decoded flow is not the rendered scene's flow (median error %.3f px)" % med
        # (2) solver stage vs the oracle chain on the device's own arrays
        r = P.solve_pair(fwd, diff, dep, depth_ref, K)
        assert out.good_kp_found == 1 and r["good_kp_found"], "the no-override path must run with good keypoints"
        assert out.n_kp == len(r["kp_ref"]) and np.array_equal(kp_ref, r["kp_ref"]) and np.array_equal(kp_cur, r["kp_cur"])
        R = np.array(out.R[:]).reshape(3, 3)
        t = np.array(out.t[:]).reshape(3, 1)
        if r["status"] == "E":
            assert out.status == 0
            assert np.array_equal(R, r["E"]["R"]) and np.array_equal(t, r["E"]["t"])
            assert np.array_equal(inl, np.asarray(r["E"]["inliers"]).reshape(-1).astype(bool))
            assert out.scale_n_valid == r["scale_diag"]["n_valid"] and abs(out.scale - r["scale"]) <= 1e-9 * abs(r["scale"])
        else:
            assert r["status"] == "PnP" and out.status == 3
            assert out.pnp_n_filtered == len(r["pnp"]["kp1"]) and out.pnp_inliers == r["pnp"]["best_inlier"]
            assert np.array_equal(R, r["pnp"]["R"]) and np.array_equal(t, r["pnp"]["t"])
        st_o, st_d = np.random.get_state(), pipe.get_rng_state()
        assert np.array_equal(st_o[1], st_d[1]) and st_o[2] == st_d[2], "numpy RandomState diverged at pair %d" % k
        rel, m = pipe.hybrid_pose(out, np.eye(4))
        assert np.abs(rel - r["pose"]).max() <= 1e-9
        modes.append(m)
        if not large:
            _, odep = P.frame_depth(dsd, seq["frames"][k + 1])
            st_dev = np.random.get_state()
            np.random.set_state(st_img)
            ri = P.solve_pair(ofwd, odiff, odep, odepth_ref, K)
            st_img = np.random.get_state()
            np.random.set_state(st_dev)
            odepth_ref = odep
            same_kp = ri["good_kp_found"] and len(ri["kp_ref"]) == len(kp_ref) and np.array_equal(ri["kp_ref"], kp_ref) \
                and np.array_equal(ri["kp_cur"], kp_cur)
            same_mask = bool(same_kp and ri["status"] == r["status"] == "E"
                             and np.array_equal(np.asarray(ri["E"]["inliers"]).reshape(-1).astype(bool), inl))
            dT = float(np.linalg.norm(rel - ri["pose"])) if ri["pose"] is not None else float("inf")
            acct.append((bool(same_kp), same_mask, dT, ri["status"] == m))
        # (3) geometry
        Tgt = np.linalg.inv(seq["poses"][k]) @ seq["poses"][k + 1]
        assert np.abs(rel[:3, :3] - Tgt[:3, :3]).max() < 2e-3
        assert np.linalg.norm(rel[:3, 3] - Tgt[:3, 3]) < (0.1 if lateral else 0.05) * np.linalg.norm(Tgt[:3, 3])
        print("%dx%d %s %s pair %d: %s kp %d inl %d | flow err vs oracle %.2e px, vs truth (median) %.3f px | |t| %.4f (gt %.4f)" % (
            h, w, mode, conv_precision, k, m, out.n_kp, out.best_inlier_cnt, e_flow, med, np.linalg.norm(rel[:3, 3]), np.linalg.norm(Tgt[:3, 3])))
        depth_ref = dep  # the current frame's depth rolls over (dfvo.py: ref_data <- cur_data)
    pipe.close()
    if mode == "mux" or lateral:
        assert "E" in modes  # the E-tracker + depth scale path is the one taken with true fwd/bwd flow / the sideways drive
    if (h, w, mode, step) == (376, 1241, "pot", 0.3):
        # forward drive under the potential encoding: bwd(p) = -fwd(p) fails the consistency check where the parallax is
        # large, the E pose is rejected and the pair goes through the PnP fallback at KITTI size (dfvo.py:225-250) -- the
        # branch the bench's ping-pong pair takes only occasionally
        assert "PnP" in modes
    if acct:
        n_kp = sum(a[0] for a in acct)
        n_mask = sum(a[1] for a in acct)
        dts = [a[2] for a in acct]
        print("FROM-IMAGES %dx%d %s %s (%s): %d pairs | identical keypoint set %d | identical inlier mask %d | same tracking "
              "mode %d | ||dT||_F <= 1e-4: %d, max %.3e, each %s" % (
                  h, w, mode, step, conv_precision, len(acct), n_kp, n_mask, sum(a[3] for a in acct),
                  sum(d <= 1e-4 for d in dts), max(dts), " ".join("%.1e" % d for d in dts)))
        assert all(a[3] for a in acct), "device and oracle-from-images took different tracking branches"
        # the nets differ from torch-CPU in fp32 summation order (<= 2e-3 px): a keypoint may cross the 0.1 px consistency
        # threshold or change rank inside a cell, after which RANSAC sees another sample stream.  The pose bar that an
        # independent from-images run CAN meet is therefore statistical; what is gated here is that it stays small
        assert max(dts) <= 5e-2 * max(1.0, float(np.linalg.norm(seq["poses"][1][:3, 3] - seq["poses"][0][:3, 3])))


def test_pipeline_config5_settings(gpu, conv_precision_any):
    conv_precision = conv_precision_any
    """BASELINE config 5 through the FUSED pipeline object exactly as `bench.py --height 1280 --width 1920 --kp-bestn 20000
    --e-max-iters 8192` drives it: 1920x1280 coded pairs, local_bestN with num_bestN 20000 (200 per cell,
    kp_selection.py:74-200), findEssentialMat with an 8192-hypothesis budget.  The solver stage is compared with the oracle
    chain (same knobs) on the device's own arrays: keypoints (values and order), inlier mask, R, t, scale and the numpy
    RandomState bit for bit, over two consecutive pairs.  Parametrised over all three net arithmetics: config 5 as BASELINE.json
    names it is the "f16" one ("fp16 flow"); the solver-stage assertions are precision-independent (the oracle chain runs on
    the device's own flow / depth arrays), the geometry gates at the end are what the f16 nets must still deliver."""
    pmod, smod = _mods()
    h, w, nb, iters = 1280, 1920, 20000, 8192
    seq = coded_tunnel_sequence(h, w, 3, mode="mux", step=1.0)
    fsd, dsd = crafted_liteflownet_state_dict(h, w, "mux"), crafted_monodepth2_state_dict()
    K = seq["K"]
    pipe = pmod.TrackingPipeline(h, w, 192, 640, K, fsd, dsd, seed=4869, kp_num_bestN=nb, e_max_iters=iters)
    fr = smod.frames_to_device(seq["frames"])
    pipe.enqueue_nets(3, fr[0], fr[0])
    pipe.sync()
    depth_ref = pipe.get_outputs(3)[4]
    pipe.set_ref_image(fr[0])
    np.random.seed(4869)
    for k in range(2):
        pipe.enqueue_nets(k, fr[k], fr[k + 1])
        if k == 1:
            pipe.prefetch_track(k)
        out = pipe.track(k)
        fwd, bwd, diff, raw, dep = pipe.get_outputs(k)
        kp_ref, kp_cur, inl = pipe.get_keypoints(k, cap=nb + 64)
        r = P.solve_pair(fwd, diff, dep, depth_ref, K, num_bestN=nb, e_max_iters=iters)
        assert out.good_kp_found == 1 and r["good_kp_found"]
        assert out.n_kp == len(r["kp_ref"]) and out.n_kp > 10000, "config 5 must really select ~20 k keypoints (got %d)" % out.n_kp
        assert np.array_equal(kp_ref, r["kp_ref"]) and np.array_equal(kp_cur, r["kp_cur"])
        R = np.array(out.R[:]).reshape(3, 3)
        t = np.array(out.t[:]).reshape(3, 1)
        assert r["status"] == "E" and out.status == 0, "the forward drive under the mux encoding is E-tracked (%s / %d)" % (r["status"], out.status)
        assert np.array_equal(R, r["E"]["R"]) and np.array_equal(t, r["E"]["t"])
        assert np.array_equal(inl, np.asarray(r["E"]["inliers"]).reshape(-1).astype(bool))
        assert out.best_inlier_cnt == r["E"]["best_inlier_cnt"]
        assert out.scale_n_valid == r["scale_diag"]["n_valid"] and abs(out.scale - r["scale"]) <= 1e-9 * abs(r["scale"])
        st_o, st_d = np.random.get_state(), pipe.get_rng_state()
        assert np.array_equal(st_o[1], st_d[1]) and st_o[2] == st_d[2], "numpy RandomState diverged at pair %d" % k
        rel, m = pipe.hybrid_pose(out, np.eye(4))
        Tgt = np.linalg.inv(seq["poses"][k]) @ seq["poses"][k + 1]
        assert np.abs(rel[:3, :3] - Tgt[:3, :3]).max() < 2e-3
        assert np.linalg.norm(rel[:3, 3] - Tgt[:3, 3]) < 0.05 * np.linalg.norm(Tgt[:3, 3])
        print("config 5 (%s) pair %d: kp %d inliers %d scale %.4f |t| %.4f (gt %.4f)" % (
            conv_precision, k, out.n_kp, out.best_inlier_cnt, out.scale, np.linalg.norm(rel[:3, 3]), np.linalg.norm(Tgt[:3, 3])))
        depth_ref = dep
    pipe.close()


def test_chunked_sequence_equals_single_chunk(gpu):
    """SURVEY 8e on one device: the sequence tracked as 1, 2 and 3 contiguous chunks (each starting from its halo frame via
    set_ref_image) gives bit-identical relative poses in the per-pair-seed mode, and the composed trajectories agree."""
    pmod, smod = _mods()
    dmod = importlib.import_module("df-vo_amd.dist")
    h, w, n = 256, 640, 9
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=11)
    pipe = pmod.TrackingPipeline(h, w, 192, 640, seq["K"], crafted_liteflownet_state_dict(h, w, "mux"),
                                 crafted_monodepth2_state_dict(), seed=4869)
    fr = smod.frames_to_device(seq["frames"])
    whole, st_whole = smod.track_chunk(pipe, fr, 0, n - 1, rng_mode="per_pair")
    assert (st_whole != 1).all()
    for world in (2, 3):
        parts = []
        for rank in range(world):
            lo, hi = dmod.chunk_bounds(n - 1, world, rank)
            rel, st = smod.track_chunk(pipe, fr, lo, hi, rng_mode="per_pair")
            parts.append(dmod.allgather_poses(rel, st, 1, 0))
        g = np.concatenate(parts, 0)
        assert np.array_equal(g[:, :16].reshape(-1, 4, 4), whole) and np.array_equal(g[:, 16], st_whole)
        traj = dmod.compose_trajectory(g)
        ref = dmod.compose_trajectory(dmod.allgather_poses(whole, st_whole, 1, 0))
        assert np.array_equal(traj, ref)
    # the sequential mode differs from the per-pair mode only through the RANSAC sampling: same geometry
    seq_rel, _ = smod.track_chunk(pipe, fr, 0, n - 1, rng_mode="sequential")
    for a, b in zip(seq_rel, whole):
        assert np.linalg.norm(a[:3, 3] - b[:3, 3]) < 0.05 * np.linalg.norm(b[:3, 3])
    pipe.close()


def test_job_of_sequences_and_the_rccl_gather(gpu, tmp_path):
    """BASELINE config 3 on one device: three sequences as one job (sequence.run_sequences) -- as 1 rank, and as the items
    of a 2- and 3-rank work list tracked one after the other on this pipeline -- give bit-identical rows, trajectories
    and files; the pose rows go through the C ABI's RCCL communicator (dfvo_comm_* / dfvo_allgather_poses, world size 1:
    the only size a 1-GPU box can run) and come back unchanged; metrics under the README's 6dof protocol."""
    pmod, smod = _mods()
    dmod = importlib.import_module("df-vo_amd.dist")
    ev = importlib.import_module("df-vo_amd.evaluation")
    h, w = 256, 640
    lens = [6, 3, 5]
    seqs, gts = [], {}
    pipe = None
    for k, n in enumerate(lens):
        seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=21 + k)
        if pipe is None:
            pipe = pmod.TrackingPipeline(h, w, 192, 640, seq["K"], crafted_liteflownet_state_dict(h, w, "mux"),
                                         crafted_monodepth2_state_dict(), seed=4869)
        seqs.append(("%02d" % k, smod.frames_to_device(seq["frames"]), n))
        gts["%02d" % k] = seq["poses"]
    comm = dmod.RcclComm(1, 0, lambda raw: raw)
    one = smod.run_sequences(pipe, seqs, 1, 0, None, comm, rng_mode="per_pair", out_dir=str(tmp_path), gts=gts, alignment="6dof",
                             compose="device")
    total = sum(n - 1 for n in lens)
    rows = np.concatenate([one[k]["gathered"] for k in one], 0)
    assert rows.shape == (total, 17) and (rows[:, 16] != 2).all()
    back = comm.allgather_rows(rows, [total])  # ncclAllGather of the whole job's rows, world 1
    assert np.array_equal(back, rows)
    for world in (2, 3):
        items = dmod.job_items([n - 1 for n in lens], world)
        parts = []
        for rank in range(world):
            for s, lo, hi in items[rank]:
                rel, st = smod.track_chunk(pipe, seqs[s][1], lo, hi, rng_mode="per_pair")
                parts.append(dmod.pack_rows(rel, st))
        assert np.array_equal(np.concatenate(parts, 0), rows)
    for name, v in one.items():
        assert np.array_equal(ev.load_traj(str(tmp_path / (name + ".txt"))), v["poses"])
        assert np.abs(v["poses"] - dmod.compose_trajectory(v["gathered"])).max() <= 1e-12
        m = v["metrics"]
        assert m["ate"] < 0.1 and m["rpe_t"] < 0.1, (name, m)  # the coded world is tracked to a few cm
        print("sequence %s: %d frames, ATE(6dof) %.4f m, RPE %.4f m / %.4f deg" % (name, len(v["poses"]), m["ate"], m["rpe_t"], m["rpe_r"]))
    comm.close()
    pipe.close()
