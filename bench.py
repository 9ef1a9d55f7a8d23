#!/usr/bin/env python
"""bench.py -- KITTI-sized frame pairs per second through the DF-VO tracking hot path on MI355X.

One step = one frame pair (1241x376): monodepth2 depth + LiteFlowNet forward/backward flow + consistency
(HIP nets: f16x3 split products on the f16 matrix cores by default, fp32-class; `exact_fp32` in the same JSON line is the
all-fp32-MFMA path timed in the same run), local_bestN keypoint selection, homography + 5 x five-point RANSAC + GRIC +
recoverPose, depth-ratio scale RANSAC (PnP fallback where the reference takes it), pose out.  Inputs (the two uint8
frames) are resident in HBM before the timed region; the Pillow-exact LANCZOS resize of the depth input runs on the
device inside it.

--solver-inputs nets (default): the product data path -- the solver stage consumes the nets' own flow / consistency /
depth.  The two frames come from the coded tunnel world (df-vo_amd/synthetic.py: frames that carry their flow and depth
as colour codes, decoded by a few channels of otherwise random weights), tracked ping-pong (A->B, B->A, ...) so that
every pair, including the rolled-over reference depth of the PnP fallback, is physically consistent.
--solver-inputs synthetic: round 1's mode -- random-weight nets (incoherent flow) computed in the timed region, the
solver stage fed a synthetic rigid-scene flow / consistency / depth triple of the same shape.

    python bench.py [--gpus N --steps K --warmup W]
N > 1: one process per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank; run
directly, bench.py re-launches itself under torch.distributed.run with N ranks on 127.0.0.1.  The N ranks track ONE
sequence of N x K pairs: contiguous chunk per rank from its 1-frame halo (df-vo_amd/sequence.py run_sequence), one
all-gather of the relative poses, prefix composition on every rank; rank 0 then re-tracks the whole sequence alone
(outside the timed region) and checks the gathered poses bit for bit.

--surface mirrors   times the drop-in class surface instead of the fused pipeline object: DeepModel / KeypointSampler /
                    EssTracker / PnpTracker built from weight FILES, host numpy arrays in and out, the frame loop of
                    DFVO.main, torch touching the GPU first; per-stage ms under the reference's Timer keys.
--frames host       frames arrive in pinned host memory: each pair's new frame is uploaded on a copy stream inside the
                    timed region, overlapped with the nets of the pairs in flight (default: frames resident in HBM).

Prints ONE JSON line on rank 0 (contract in the task description; `roofline` and `cpu_baseline` added).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")  # before torch / HIP initialise: see df-vo_amd/__init__.py

SLOTS = 4  # DFVO_PIPELINE_SLOTS
PREFETCH = os.environ.get("DFVO_BENCH_PREFETCH", "1") != "0"  # RNG-independent solver half enqueued behind the nets
RAMP = int(os.environ.get("DFVO_BENCH_RAMP", "1"))  # pairs whose nets are enqueued before the first chain begins (3 = rounds 1-4)
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix (f32 in / f32 acc)
PEAK_F16_MFMA_TFLOPS = 2500.0  # same table: F16 / BF16 MFMA dense
PEAK_HBM_GBS = 8000.0  # same guide: HBM3E ~8 TB/s
PMC_FILE = "r6_pmc_bench.json"  # committed rocprofv3 --pmc passes over the default command (tools/profile.sh)
STATS_FILE = "r6_rocprofv3_kernel_stats.csv"  # ... and its --kernel-trace --stats summary


def kernel_source_digest():
    """digest of the conv kernel sources: ties a committed PMC profile to the build it was taken on"""
    import hashlib
    h = hashlib.sha1()
    for f in ("conv_igemm_f32.hip", "conv_win_f16s.h", "conv_win_f16s2.h", "conv_gemm_f16s.h", "conv_epi.h", "conv_f16_split.h",
              "conv_taps_f16s.hip", "conv_gemm_f32g.hip"):
        try:
            h.update(open(os.path.join(ROOT, "df-vo_amd", "csrc", f), "rb").read())
        except OSError:
            pass
    return h.hexdigest()[:12]


CFG_NAMES = ["conv_igemm_f32<2,2,4,4> 128x128", "conv_igemm_f32<1,4,2,2> 32x128", "conv_igemm_f32<4,1,4,4> 256x64",
             "conv_igemm_f32<2,2,2,2> 64x64", "conv_igemm_f32<4,1,4,2> 256x32", "conv_igemm_f32<2,2,2,1> 64x32",
             "conv_igemm_f32<4,1,4,1> 256x16", "conv_igemm_f32<4,1,1,1> 64x16", "conv_igemm_f32<1,4,4,2> 64x128",
             "conv_igemm_f32<2,2,4,2> 128x64", "conv_igemm_f32<4,1,2,2> 128x32", "conv_igemm_f32<4,1,2,1> 128x16",
             "conv_win3_f32<2,2,4,4> (8x16)x128", "conv_win3_f32<2,2,4,2> (8x16)x64", "conv_win3_f32<4,1,2,2> (8x16)x32",
             "conv_win3_f32<1,4,4,2> (4x16)x128", "conv_head_f32<7> 7x7 heads (direct)",
             "conv_head_f32<5> 5x5 heads (direct)", "conv_head_f32<3> 3x3 heads (direct)",
             "conv_win_f16s (4|8 x 32) x 128|64|32 (f16 hi/lo planes, v_mfma_f32_32x32x16_f16)",
             "f16x3 streaming layers (conv_gemm_f16s <4,1,*>: 1x1 / stride 2; conv_taps_f16s: 7x7, k x 1, 5x5 from an LDS window; f16 hi/lo planes)",
             "conv_gemm_f16s K-sliced (small maps: pyramid levels 5-6, depth net inner layers; in-workgroup ordered reduction)",
             "conv_gemm_f32g streaming (exact fp32 mode: 1x1, k x 1, stride 2, 7x7 layers; v_mfma_f32_32x32x2_f32, register ring)",
             "conv_gemm_f32g K-sliced (exact fp32 mode: small maps; in-workgroup ordered reduction)"]


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_nets(frames, fsd, dsd, K, H, W, n_pairs=3, warmup=None):
    """the oracle's image-level frame loop (oracle/pipeline_np.py's stages: torch-CPU nets on all host cores + single-threaded
    C/numpy solvers, as OpenCV's RANSAC is) over the same coded frames, ping-pong like the timed region.  BASELINE.md
    section 4's protocol: warm-up pairs (3 when n_pairs >= 20, else 1), then the mean over n_pairs, per-stage ms under the
    reference's Timer keys (libs/general/timer.py; dfvo.py:146-246,305-335), core count and CPU model stated."""
    import torch
    from oracle import nets_torch as O
    from oracle import pipeline_np as P
    from oracle import tracker_np as T
    cores = torch.get_num_threads()
    if warmup is None:
        warmup = 3 if n_pairs >= 20 else 1
    keys = ["depth_cnn", "flow_cnn", "kp_sel", "E-tracker", "scale_recovery", "pnp"]
    acc = {k: 0.0 for k in keys}
    status = []
    np.random.seed(4869)  # apis/run.py:81-84
    _, depth_ref = P.frame_depth(dsd, frames[0])
    t_begin = None
    for k in range(1, warmup + n_pairs + 1):
        if k == warmup + 1:
            acc = {kk: 0.0 for kk in keys}
            status = []
            t_begin = time.perf_counter()
        ref, cur = frames[(k - 1) % 2], frames[k % 2]
        t0 = time.perf_counter()
        _, depth_cur = P.frame_depth(dsd, cur)
        t1 = time.perf_counter()
        fwd, bwd, diff = O.flow_inference(fsd, ref, cur)
        t2 = time.perf_counter()
        kp = T.local_bestN(fwd, diff[..., None] if diff.ndim == 2 else diff)
        t3 = time.perf_counter()
        acc["depth_cnn"] += t1 - t0
        acc["flow_cnn"] += t2 - t1
        acc["kp_sel"] += t3 - t2
        mode = "constant_motion"
        if kp["good_kp_found"]:
            kp1, kp2 = kp["kp1_best"][0], kp["kp2_best"][0]
            res = T.compute_pose_2d2d(kp1, kp2, K)
            t4 = time.perf_counter()
            acc["E-tracker"] += t4 - t3
            scale = -1
            if np.linalg.norm(res["t"]) != 0:
                pose = np.eye(4)
                pose[:3, :3], pose[:3, 3:] = res["R"], res["t"]
                scale = T.find_scale_from_depth(kp1, kp2, np.linalg.inv(pose), depth_cur, K)
            t5 = time.perf_counter()
            acc["scale_recovery"] += t5 - t4
            mode = "E"
            if np.linalg.norm(res["t"]) == 0 or scale == -1:
                T.compute_pose_3d2d(kp1, kp2, depth_ref, K)
                acc["pnp"] += time.perf_counter() - t5
                mode = "PnP"
        status.append(mode)
        depth_ref = depth_cur
    dt = time.perf_counter() - t_begin
    return {"value": n_pairs / dt, "unit": "frames/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model(),
            "stage_ms_per_pair": {kk: round(v / n_pairs * 1e3, 1) for kk, v in acc.items()},
            "sample": "%d frame pairs %dx%d after %d warm-up pair(s) through the oracle frame loop (torch-CPU fp32 monodepth2 + "
                      "LiteFlowNet fwd+bwd on %d threads, single-threaded C/numpy solvers on their outputs; tracked %s), %.1f s" % (
                          n_pairs, W, H, warmup, cores, "/".join(status), dt)}


def cpu_baseline(syn, H, W, scenes, n_pairs=2):
    """the oracle (torch-CPU nets + C/numpy solvers) timed on the host cores for a bounded sample"""
    import torch
    from PIL import Image
    from oracle import nets_torch as O
    from oracle import tracker_np as T
    fsd, dsd = syn.liteflownet_state_dict(4869), syn.monodepth2_state_dict(4869)
    ref, cur = syn.image_pair(H, W, seed=1)
    np.random.seed(4869)
    cores = torch.get_num_threads()
    t0 = time.time()
    for i in range(n_pairs):
        feed = np.asarray(Image.fromarray(cur).resize((640, 192), Image.LANCZOS))  # deep_models.py:195-199
        depth = O.depth_inference(dsd, feed)
        fwd, bwd, diff = O.flow_inference(fsd, ref, cur)
        sc = scenes[i % len(scenes)]
        kp = T.local_bestN(sc["flow"], sc["diff"][..., None])
        res = T.compute_pose_2d2d(kp["kp1_best"][0], kp["kp2_best"][0], sc["K"])
        scale = -1
        if np.linalg.norm(res["t"]) != 0:
            pose = np.eye(4)
            pose[:3, :3], pose[:3, 3:] = res["R"], res["t"]
            scale = T.find_scale_from_depth(kp["kp1_best"][0], kp["kp2_best"][0], np.linalg.inv(pose), sc["depth_cur"],
                                            sc["K"])
        if np.linalg.norm(res["t"]) == 0 or scale == -1:  # PnP fallback (dfvo.py:225-250)
            T.compute_pose_3d2d(kp["kp1_best"][0], kp["kp2_best"][0], sc["depth_ref"], sc["K"])
    dt = time.time() - t0
    return {"value": n_pairs / dt, "unit": "frames/s", "cores": int(cores), "kind": "port",
            "sample": "%d frame pairs 1241x376: torch-CPU fp32 monodepth2 + LiteFlowNet (fwd+bwd) and the C/numpy "
                      "solver oracle on the same synthetic inputs, %.1f s" % (n_pairs, dt)}


def spawn_ranks(n):
    """`python bench.py --gpus N` run directly: re-launch under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def world_setup(args, torch):
    """(rank, world, local_rank, dist or None); backend nccl (= RCCL) on GPUs, gloo when DFVO_BENCH_BACKEND=gloo (CPU tests)"""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("DFVO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank, dist


def ranks_seen(dist, world, rank, local_rank, torch):
    """distinct (rank, device) pairs that took part, from an all-gather"""
    if torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(local_rank if world > 1 else 0)
        me = (rank, local_rank if world > 1 else 0, getattr(pr, "uuid", None) and str(pr.uuid), getattr(pr, "pci_bus_id", None))
    else:
        me = (rank, -1, None, None)
    if dist is None:
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me)
    return out


def to_device(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def device_sync():
    import torch
    torch.cuda.synchronize()


def run_mirrors(args, syn, capi, h_frames, K, fsd, dsd, code_mode):
    """--surface mirrors: the frame loop of DFVO.main (/root/reference/libs/dfvo.py:347-425: deep_model_inference :299-345,
    tracking :121-262, update_global_pose :109-119) over the drop-in classes exactly as libs/dfvo.py instantiates them --
    DeepModel(cfg).initialize_models() from weight FILES in the reference's on-disk formats, KeypointSampler(cfg),
    EssTracker / PnpTracker(cfg, cam_intrinsics) -- with host numpy arrays in and out of every call (2 x 1.4 MB H2D +
    9.8 MB D2H per pair, every call blocking), the nets in --conv-precision (DeepModel.initialize_models reads it from the
    environment), the frame session of df-vo_amd/libs/deep_models/session.py unless DFVO_SESSION=0, and torch touching the GPU
    before the library creates its streams (as apis/run.py's imports do).  Per-stage host times
    under the reference's Timer keys (libs/general/timer.py; dfvo.py:146-246,305-335)."""
    import tempfile
    import torch
    torch.zeros(8, device="cuda").sum().item()  # the caller used the GPU first
    cfg_mod = importlib.import_module("df-vo_amd.default_cfg")
    dm_mod = importlib.import_module("df-vo_amd.libs.deep_models.deep_models")
    cam_mod = importlib.import_module("df-vo_amd.libs.geometry.camera_modules")
    ks_mod = importlib.import_module("df-vo_amd.libs.matching.keypoint_sampler")
    trk_mod = importlib.import_module("df-vo_amd.libs.tracker")
    H, W = args.height, args.width
    tmp = tempfile.mkdtemp(prefix="dfvo_bench_")
    flow_path, depth_dir = syn.write_weight_files(tmp, fsd, dsd)
    cfg = cfg_mod.default_configuration(H, W, flow_path, depth_dir)
    deep_models = dm_mod.DeepModel(cfg)
    deep_models.initialize_models()
    cam = cam_mod.Intrinsics([K[0, 2], K[1, 2], K[0, 0], K[1, 1]])
    sampler = ks_mod.KeypointSampler(cfg)
    e_tracker, pnp_tracker = trk_mod.EssTracker(cfg, cam, None), trk_mod.PnpTracker(cfg, cam)
    SE3 = cam_mod.SE3
    np.random.seed(cfg.seed)
    keys = ["depth_cnn", "flow_cnn", "kp_sel", "E-tracker", "scale_recovery", "pnp", "deep_inference", "tracking", "DF-VO"]
    acc = {k: 0.0 for k in keys}
    fh, fw = deep_models.depth.feed_height, deep_models.depth.feed_width
    ys = np.minimum(np.floor(np.arange(H) * (fh / float(H))).astype(np.int64), fh - 1)  # cv2.resize INTER_NEAREST, dfvo.py:314-317
    xs = np.minimum(np.floor(np.arange(W) * (fw / float(W))).astype(np.int64), fw - 1)
    y0, y1 = int(H * cfg.crop.depth_crop[0][0]), int(H * cfg.crop.depth_crop[0][1])
    x0, x1 = int(W * cfg.crop.depth_crop[1][0]), int(W * cfg.crop.depth_crop[1][1])
    crop_mask = np.zeros((H, W))
    crop_mask[y0:y1, x0:x1] = 1
    ref_data, cur_data = {}, {}
    global_pose = SE3()
    modes = []
    n_frames = args.warmup + args.steps + 1
    t_begin = None
    for img_id in range(n_frames):
        if img_id == args.warmup + 1:  # frame 0 only gets its depth; `warmup` untimed pairs, then `steps` timed ones
            acc = {k: 0.0 for k in keys}
            modes = []
            t_begin = time.perf_counter()
        tf0 = time.perf_counter()
        cur_data["id"], cur_data["timestamp"], cur_data["img"] = img_id, img_id, h_frames[img_id % 2]
        t0 = time.perf_counter()
        raw = deep_models.forward_depth(imgs=[cur_data["img"]])
        cur_data["raw_depth"] = raw[ys][:, xs]
        d = cur_data["raw_depth"]
        cur_data["depth"] = d * (crop_mask * ((d < cfg.depth.max_depth) * (d > cfg.depth.min_depth)))  # preprocess_depth, utils.py:89-114
        t1 = time.perf_counter()
        acc["depth_cnn"] += t1 - t0
        if img_id >= 1:
            flows = deep_models.forward_flow(cur_data, ref_data, forward_backward=cfg.deep_flow.forward_backward)
            ref_data["flow"] = flows[(ref_data["id"], cur_data["id"])].copy()        # dfvo.py:330-333
            cur_data["flow"] = flows[(cur_data["id"], ref_data["id"])].copy()
            ref_data["flow_diff"] = flows[(ref_data["id"], cur_data["id"], "diff")].copy()
            t2 = time.perf_counter()
            acc["flow_cnn"] += t2 - t1
            acc["deep_inference"] += t2 - t0
            kp_sel = sampler.kp_selection(cur_data, ref_data)
            t3 = time.perf_counter()
            acc["kp_sel"] += t3 - t2
            mode = "Constant motion"
            hybrid = SE3()
            if kp_sel["good_kp_found"]:
                sampler.update_kp_data(cur_data, ref_data, kp_sel)
                mode = "Ess. Mat."
                e_out = e_tracker.compute_pose_2d2d(ref_data["kp_best"], cur_data["kp_best"], True)
                E_pose = e_out["pose"]
                hybrid.R = E_pose.R
                t4 = time.perf_counter()
                acc["E-tracker"] += t4 - t3
                scale = -1
                if np.linalg.norm(E_pose.t) != 0:
                    scale = e_tracker.scale_recovery(cur_data, ref_data, E_pose, False)["scale"]
                    if scale != -1:
                        hybrid.t = E_pose.t * scale
                t5 = time.perf_counter()
                acc["scale_recovery"] += t5 - t4
                if np.linalg.norm(E_pose.t) == 0 or scale == -1:
                    hybrid = pnp_tracker.compute_pose_3d2d(ref_data["kp_best"], cur_data["kp_best"], ref_data["depth"], True)["pose"]
                    mode = "PnP"
                    acc["pnp"] += time.perf_counter() - t5
            global_pose.t = global_pose.R @ hybrid.t + global_pose.t
            global_pose.R = global_pose.R @ hybrid.R
            acc["tracking"] += time.perf_counter() - t2
            modes.append(mode)
        else:
            acc["deep_inference"] += t1 - t0
        ref_data = dict(cur_data)
        ref_data["flow"] = cur_data["flow"] = ref_data["flow_diff"] = None
        acc["DF-VO"] += time.perf_counter() - tf0
    capi.check(capi.lib().dfvo_sync_device())
    dt = time.perf_counter() - t_begin
    net_h, net_w = syn._net_size(H, W)
    line = {
        "metric": "KITTI-odom frames/sec (DF-VO per-pair tracking hot path: monodepth2 + LiteFlowNet fwd/bwd + "
                  "kp selection + E/H RANSAC + scale)",
        "value": round(args.steps / dt, 3), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "f16x3": "f32 (f16x3 split products)", "f16": "f16 (one product per term, fp32 accumulate)"}[args.conv_precision], "data": "synthetic",
        "config": {"workload": "%dx%d frame pairs (flow net %dx%d batch 2; depth net 192x640) through the drop-in class surface: "
                               "DeepModel.forward_depth / forward_flow, KeypointSampler.kp_selection, EssTracker.compute_pose_2d2d / "
                               "scale_recovery, PnpTracker.compute_pose_3d2d built from weight files; host numpy arrays in and out of "
                               "every call, synchronous; torch touched the GPU first" % (W, H, net_h, net_w),
                   "surface": "mirrors", "conv_precision": args.conv_precision,
                   "solver_inputs": "the nets' own outputs, coded tunnel-world frames, '%s' encoding, ping-pong" % code_mode,
                   "tracked_by_E": modes.count("Ess. Mat."), "tracked_by_PnP": modes.count("PnP"),
                   "constant_motion": modes.count("Constant motion")},
        "stage_ms_per_pair": {k: round(v / args.steps * 1e3, 3) for k, v in acc.items()},
        "roofline": None, "cpu_baseline": None,
        "session": dict(deep_models.session.stats) if getattr(deep_models, "session", None) is not None else None}
    print(json.dumps(line))


def committed_streaming_kernels():
    """the non-conv streaming kernels (correlation, warp, depth-wise deconvolution, resizes, flow mean ...) against the HBM
    roofline, from the committed rocprofv3 passes over the default command: bytes per dispatch (PMC: FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE, tools/pmc_traffic.py) over the average duration of the same kernel in the kernel-trace statistics"""
    import csv
    import re

    def norm(name):  # "void dfvo::k_correlation_rt<64, 1>(float const*, ...)" -> "k_correlation_rt"
        n = re.sub(r"<.*", "", name.split("(")[0].replace("void ", "").strip())
        return n.split("::")[-1]
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))["kernels"]
        dur = {}
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", STATS_FILE))):
            t = dur.setdefault(norm(r["Name"]), [0.0, 0])
            t[0] += float(r["TotalDurationNs"])
            t[1] += int(r["Calls"])
        byt = {}
        for k, v in prof.items():
            t = byt.setdefault(norm(k), [0.0, 0])
            t[0] += v["hbm_bytes_per_dispatch"] * v["dispatches"]
            t[1] += v["dispatches"]
    except (OSError, KeyError, ValueError):
        return None
    out = []
    # the nets' own non-conv operators (the solver stage's kernels are latency chains of dependent fp64 / integer work, not streams)
    streaming = ("k_correlation", "k_warp", "k_deconv", "k_reg_head", "k_reg_prep", "k_flow_mean", "k_maxpool", "k_copy_segments",
                 "k_resize_bilinear", "k_flow_resize", "k_flow_consistency", "k_img_u8", "k_disp_to_depth", "k_depth_post", "k_lanczos")
    for k, (b, nd) in byt.items():
        if not k.startswith(streaming) or k not in dur or nd == 0 or b / nd < 4e6:
            continue  # (conv kernels are priced live; small kernels are launch-latency bound, not streaming)
        ns = dur[k][0] / dur[k][1]
        gbs = b / nd / ns
        e = {"kernel": k, "avg_us": round(ns / 1e3, 2), "mb_per_dispatch": round(b / nd / 1e6, 2),
             "achieved_gbs": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4), "calls": dur[k][1]}
        if ns < 10e3:
            e["latency_sized"] = True  # under 10 us a launch is its ~3.4 us of dispatch + one memory round trip: frac says little
        out.append(e)
    return sorted(out, key=lambda e: -e["avg_us"] * e["calls"])[:8] or None


def gpu_visible():
    """is there a GPU, asked WITHOUT creating a HIP context in this process: the KFD device node (with its sysfs topology as
    a second opinion when it is readable)"""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        nodes = "/sys/class/kfd/kfd/topology/nodes"
        simd = []
        for n in os.listdir(nodes):
            with open(os.path.join(nodes, n, "properties")) as f:
                simd += [int(l.split()[1]) for l in f if l.startswith("simd_count")]
        return any(v > 0 for v in simd) if simd else True
    except (OSError, ValueError):
        return True


def other_leg(argv, timeout=300):
    """one short leg of the default line: bench.py itself in a child process, its JSON line reduced to the figures"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-roofline", "--no-exact-leg", "--no-other-legs"]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:  # reported in the line, never fatal for the headline
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200]), "argv": " ".join(argv)}
    out = {"value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
           "dtype": d["dtype"].split(":")[0].split(" (")[0] if len(d["dtype"]) > 40 else d["dtype"], "argv": " ".join(argv),
           "workload": d["config"]["workload"][:160], "leg_wall_s": round(time.perf_counter() - t0, 1)}
    for k in ("tracked_by_E", "tracked_by_PnP", "constant_motion", "conv_precision", "sequences", "frames"):
        if k in d["config"]:
            out[k] = d["config"][k]
    for k in ("stage_ms_per_pair", "session", "steady_state"):
        if d.get(k) is not None:
            out[k] = d[k]
    return out


KITTI_FRAMES = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]  # dataset/kitti_odom/gt_poses/00..10.txt


def run_job(args, pipe, d_frames, dist, world, rank, local_rank, on_gpu, torch, syn, code_mode):
    """--sequences kitti-lengths: BASELINE config 3.  Eleven sequences (frame counts of KITTI 00-10 x --scale, ping-pong coded
    frames) as one job over `world` ranks: sequence.run_sequences -- balanced (sequence, chunk) work list, 1-frame halo per
    item, ONE all-gather of pose rows (through the C ABI's RCCL communicator on GPUs), per-sequence composition, eleven
    trajectories.  Timed: the whole job incl. gather and composition, max over ranks."""
    smod = importlib.import_module("df-vo_amd.sequence")
    dmod = importlib.import_module("df-vo_amd.dist")

    class PingPong:
        def __getitem__(self, i):
            return d_frames[i % 2]

    frames = PingPong()
    seqs = [("%02d" % k, frames, max(2, int(round(n * args.scale)))) for k, n in enumerate(KITTI_FRAMES)]
    n_pairs = sum(n - 1 for _, _, n in seqs)
    comm = None
    if dist is not None and dist.get_backend() == "nccl":
        comm = dmod.RcclComm.from_torch(dist, world, rank)  # dfvo_comm_* / dfvo_allgather_poses: the C ABI's collective
    rng_mode = "sequential" if world == 1 else "per_pair"
    smod.track_chunk(pipe, frames, 0, 5, rng_mode="per_pair")  # warm-up (clocks, autotuned layer configurations)
    if dist is not None:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    res = smod.run_sequences(pipe, seqs, world, rank, dist, comm, rng_mode=rng_mode)
    device_sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    status = np.concatenate([v["gathered"][:, 16] for v in res.values()]).astype(np.int64)
    check = None
    if rank == 0 and world > 1 and on_gpu:  # outside the timed region: the same job on this rank alone, per-pair RandomState
        one = smod.run_sequences(pipe, seqs, 1, 0, None, rng_mode="per_pair")
        same = all(np.array_equal(one[k]["gathered"][:, 16], res[k]["gathered"][:, 16]) and
                   np.array_equal(one[k]["gathered"][one[k]["gathered"][:, 16] != 1], res[k]["gathered"][one[k]["gathered"][:, 16] != 1])
                   for k in res)
        check = {"sequences": len(seqs), "pairs": int(n_pairs), "equal_to_single_rank_run": bool(same)}
    seen = ranks_seen(dist, world, rank, local_rank, torch)
    if comm is not None:
        comm.close()
    pipe.close()
    if rank == 0:
        net_h, net_w = syn._net_size(args.height, args.width)
        items = dmod.job_items([n - 1 for _, _, n in seqs], world)
        line = {
            "metric": "KITTI-odom frames/sec (DF-VO per-pair tracking hot path: monodepth2 + LiteFlowNet fwd/bwd + "
                      "kp selection + E/H RANSAC + scale)",
            "value": round(n_pairs / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": int(n_pairs), "warmup": 5,
            "ms_per_step": round(dt / n_pairs * 1e3 * world, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (%s)" % args.conv_precision, "data": "synthetic",
            "config": {"workload": "BASELINE config 3: eleven sequences with the frame counts of KITTI 00-10 x %g = %s (%d pairs), %dx%d "
                                   "coded frames (flow net %dx%d), tracked as one job frame-batched over %d rank(s)" % (
                                       args.scale, [n for _, _, n in seqs], n_pairs, args.width, args.height, net_h, net_w, world),
                       "parallelism": "balanced (sequence, chunk) work list, 1-frame halo per item, %s RandomState, ONE all-gather of pose rows%s, "
                                      "per-sequence prefix composition" % (rng_mode, " (dfvo_allgather_poses: RCCL ncclAllGather)" if comm else ""),
                       "items_per_rank": [len(it) for it in items], "pairs_per_rank": [sum(h - l for _, l, h in it) for it in items],
                       "tracked_by_E": int((status == 0).sum()), "tracked_by_PnP": int((status == 3).sum()),
                       "constant_motion": int((status == 1).sum()), "trajectories": len(res),
                       "ranks_seen": len(set(r[0] for r in seen)), "devices_seen": len(set((r[1], r[2], r[3]) for r in seen))},
            "sequence_check": check, "roofline": None, "cpu_baseline": None}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-precision", default=os.environ.get("DFVO_CONV_PRECISION", "f16x3"), choices=["fp32", "f16x3", "f16"],
                    help="arithmetic of the conv layers: f16x3 (default: fp32-class split kernels, the whole -m gpu net / "
                         "pipeline parity suite runs green on it at the exact path's tolerances), fp32 (exact fp32 MFMA), "
                         "f16 (one f16 product per term, fp32 accumulate: BASELINE config 5's 'fp16 flow' -- NOT fp32-class, "
                         "reported under its own dtype) -- reported in dtype and config.conv_precision")
    ap.add_argument("--solver-inputs", default="nets", choices=["nets", "synthetic"],
                    help="nets: the solver stage consumes the nets' own outputs (coded-world frames, the product data path); "
                         "synthetic: random-weight nets + a synthetic rigid-scene flow/consistency/depth triple (round-1 mode)")
    ap.add_argument("--e-max-iters", type=int, default=1000, help="findEssentialMat hypothesis budget (config 5: 8192)")
    ap.add_argument("--kp-bestn", type=int, default=2000, help="kp_selection.local_bestN.num_bestN (config 5: 20000)")
    ap.add_argument("--surface", default="fused", choices=["fused", "mirrors"],
                    help="fused: the dfvo_pipeline_* object (default); mirrors: the reference's class surface over host arrays")
    ap.add_argument("--frames", default="device", choices=["device", "host"],
                    help="host: pinned host frames, H2D on a copy stream inside the timed region")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the second timed leg in exact fp32 (exact_fp32 in the JSON line)")
    ap.add_argument("--no-other-legs", action="store_true",
                    help="skip the short extra legs of the default line: dropin_surface (the reference's class surface, 8 pairs), "
                         "other_configs (BASELINE configs 3 / 4 / 5, a few pairs each) and features_recomputed")
    ap.add_argument("--feature-carry", default="on", choices=["on", "off"],
                    help="on: the flow net's image / feature pyramids of a pair's reference frame are the ones the previous pair "
                         "computed for it as its current frame (one Features pass per new frame); off: both frames of every pair "
                         "run through Features, as the reference's model call does")
    ap.add_argument("--sequences", default=None, choices=["kitti-lengths"],
                    help="BASELINE config 3: eleven synthetic sequences with the frame counts of KITTI 00-10 (x --scale) tracked as "
                         "ONE job frame-batched over the ranks (df-vo_amd/sequence.py run_sequences): balanced work list, 1-frame "
                         "halo per item, one all-gather, per-sequence composition; --steps / --warmup are ignored")
    ap.add_argument("--scale", type=float, default=0.02, help="--sequences: fraction of the 23 201 KITTI frames (>= 2 frames per sequence)")
    ap.add_argument("--cpu-pairs", type=int, default=3,
                    help="pairs of the cpu_baseline leg (default 3: ~17 s on the GPU box's host cores; BASELINE.md section 4's protocol "
                         "is >= 20 pairs after 3 warm-up pairs: --cpu-pairs 20 takes ~2.5 min)")
    args = ap.parse_args(argv)
    os.environ["DFVO_CONV_PRECISION"] = args.conv_precision  # read once by the library when the layers are packed
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # Short GPU-only legs of the default line, so that the driver's ONE command also observes (a) the surface the reference's
    # own frame loop calls and (b) the other BASELINE configurations.  Each is this script again in a child process, run
    # BEFORE this process creates its HIP context: a child that shares the GPU with a parent holding twelve hardware queues
    # measured 77 frames/s through the class surface against 132 alone (profiles/r5c_*: queue over-subscription), and none of
    # it overlaps the headline's timed region.
    dropin = others = frames_host = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_other_legs and args.surface == "fused" and not args.sequences
            and args.solver_inputs == "nets" and (args.height, args.width) == (376, 1241) and args.kp_bestn == 2000
            and args.e_max_iters == 1000 and args.frames == "device" and gpu_visible()):
        dropin = other_leg(["--surface", "mirrors", "--steps", "30", "--warmup", "5", "--conv-precision", args.conv_precision])
        # SURVEY 8(d) defines the metric as images in -> pose out; `value` is, by the bench contract, the rate with the frames
        # resident in HBM -- this leg is the same run with every new frame arriving from pinned host memory inside the timed
        # region (1.4 MB per frame over PCIe on a copy stream, overlapped with the nets of the pairs in flight)
        frames_host = other_leg(["--frames", "host", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                 "--conv-precision", args.conv_precision])
        big = ["--height", "1280", "--width", "1920", "--kp-bestn", "20000", "--e-max-iters", "8192", "--steps", "5", "--warmup", "2"]
        others = {
            "config3_kitti_00_10_job": other_leg(["--sequences", "kitti-lengths", "--scale", "0.005", "--conv-precision", args.conv_precision]),
            "config4_1280x960": other_leg(["--height", "960", "--width", "1280", "--steps", "5", "--warmup", "2",
                                           "--conv-precision", args.conv_precision]),
            "config5_1920x1280_8192hyp_20kkp_f16": other_leg(big + ["--conv-precision", "f16"]),
            "config5_1920x1280_8192hyp_20kkp_" + args.conv_precision: other_leg(big + ["--conv-precision", args.conv_precision]),
        }

    import torch
    rank, world, local_rank, dist = world_setup(args, torch)
    on_gpu = dist is None or dist.get_backend() == "nccl"  # (gloo: CPU test of the multi-rank logic, tests/test_bench_cpu.py)
    # DFVO_BENCH_ONE_DEVICE=1 (verification aid for 1-GPU boxes): every rank drives device 0 with its own pipeline, collectives
    # over gloo on host tensors -- the chunk / halo / gather / compose logic and the single-rank equality check run on real
    # pipelines; only RCCL itself is left out
    one_device = world > 1 and os.environ.get("DFVO_BENCH_ONE_DEVICE") == "1" and torch.cuda.is_available()
    if one_device:
        on_gpu = True
        local_rank = 0
    if world == 1 or one_device:
        torch.cuda.set_device(0)

    pkg = importlib.import_module("df-vo_amd")  # noqa: F841
    capi = importlib.import_module("df-vo_amd.capi")
    syn = importlib.import_module("df-vo_amd.synthetic")
    pmod = importlib.import_module("df-vo_amd.pipeline")
    dmod = importlib.import_module("df-vo_amd.dist")
    if on_gpu:
        capi.check(capi.lib().dfvo_set_device(local_rank if world > 1 else 0))

    H, W = args.height, args.width
    nets_mode = args.solver_inputs == "nets"
    smod = importlib.import_module("df-vo_amd.sequence") if world > 1 else None
    dev = to_device
    d_feed = None  # the pipeline resizes the current frame itself (device LANCZOS, bit-exact with Pillow)
    popts = dict(seed=4869, e_max_iters=args.e_max_iters, kp_num_bestN=args.kp_bestn)
    # Everything the host prepares (frames, scenes, weights) first, THEN the pipeline, THEN the device copies of the inputs:
    # the pair rate depends on the order in which the process creates its HIP streams (133 vs 109 frames/s, exact fp32, same
    # binary and box: which compute pipes the pipeline's streams share follows their creation order, see
    # TrackerBuffers::init and DESIGN.md section 5)
    if nets_mode:
        # coded tunnel world: multiplexed encoding when the frame needs no input resize, potential encoding otherwise
        code_mode = "mux" if syn._net_size(H, W) == (H, W) else "pot"
        # (potential encoding: the sideways drive, for which it yields E-tracked pairs -- tunnel_poses_lateral)
        seq = syn.coded_tunnel_sequence(H, W, 2, mode=code_mode, step=1.0, seed=7,
                                        poses=None if code_mode == "mux" else syn.tunnel_poses_lateral(2, 0.4))
        K = seq["K"]
        fsd, dsd = syn.crafted_liteflownet_state_dict(H, W, code_mode), syn.crafted_monodepth2_state_dict()
        scenes = None
        h_frames = [seq["frames"][0], seq["frames"][1]]
    else:
        if world > 1:
            raise SystemExit("bench.py: --gpus N tracks one sequence through the product data path (--solver-inputs nets)")
        scenes = [syn.rigid_scene(H, W, seed=100 + i) for i in range(4)]
        K = scenes[0]["K"]
        fsd, dsd = syn.liteflownet_state_dict(4869), syn.monodepth2_state_dict(4869)
        h_frames = list(syn.image_pair(H, W, seed=1))

    if args.surface == "mirrors":
        if world > 1 or not nets_mode:
            raise SystemExit("bench.py: --surface mirrors is a single-GPU mode over the product data path")
        return run_mirrors(args, syn, capi, h_frames, K, fsd, dsd, code_mode)

    if os.environ.get("DFVO_BENCH_TORCH_FIRST") and on_gpu:  # (A/B aid: the caller used the GPU through torch before the library)
        torch.zeros(1 << 20, device="cuda").sum().item()
    pipe = pmod.TrackingPipeline(H, W, 192, 640, K, fsd, dsd, **popts)
    d_frames = [dev(f) for f in h_frames]
    if not nets_mode:
        d_sc = [(dev(s["flow"]), dev(s["diff"]), dev(s["depth_cur"])) for s in scenes]
        d_ref_depth = dev(scenes[0]["depth_ref"])
    if args.sequences:
        if not nets_mode:
            raise SystemExit("bench.py: --sequences runs the product data path (--solver-inputs nets)")
        return run_job(args, pipe, d_frames, dist, world, rank, local_rank, on_gpu, torch, syn, code_mode)
    host_frames = args.frames == "host" and on_gpu
    if host_frames:
        # frames arrive in pinned host memory; the new frame of every pair is uploaded on a copy stream while the nets of
        # the pairs in flight run.  Ring of device buffers: a buffer is rewritten SLOTS + 3 frames later, long after the
        # nets of the last pair that read it have finished (track() of that pair has returned before the upload is issued)
        ring = [torch.empty_like(d_frames[0]) for _ in range(SLOTS + 3)]
        h_pinned = [torch.from_numpy(np.ascontiguousarray(f)).pin_memory() for f in h_frames]
        copy_stream = torch.cuda.Stream()
        pending = {}

        def upload(fidx):
            buf = ring[fidx % len(ring)]
            with torch.cuda.stream(copy_stream):
                buf.copy_(h_pinned[fidx % 2], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            pending[fidx] = ev

        def frame_on_device(fidx):  # ordering contract of enqueue_nets: the frame is complete when it is handed over
            ev = pending.pop(fidx, None)
            if ev is not None:
                ev.synchronize()
            return ring[fidx % len(ring)]

    host_t = [0.0, 0.0]  # host seconds inside enqueue_nets / track (DFVO_BENCH_TRACE=1 prints them)
    t_track = []         # completion time of every track() of the last run (steady-state rate)

    carry = [args.feature_carry == "on"]

    def run(p, n):
        """software pipeline: the nets of pairs k+1, k+2 are enqueued before the solver stage of pair k blocks the host"""
        g = np.eye(4)
        prev = np.eye(4)
        rel_all = np.zeros((n, 4, 4))
        status = np.zeros(n, np.int64)
        del t_track[:]
        if n == 0:
            return rel_all, status
        if host_frames:
            pending.clear()
            upload(0)
            upload(1)
            p.set_ref_image(frame_on_device(0))
        elif nets_mode:
            p.set_ref_image(d_frames[0])  # depth of the first reference frame (PnP fallback input), from the frame itself
        else:
            p.set_ref_depth(depth=d_ref_depth)
        # nets run `ahead` pairs ahead of the solver stage (<= SLOTS - 1): with 3, each of the two flow-net instances always
        # has its next pair queued behind the current one while track(k) blocks the host
        ahead = int(os.environ.get("DFVO_BENCH_AHEAD", "3"))

        def feed(j):  # nets of pair j, then the RNG-independent half of its solver stage right behind them
            # carried mode: from the second pair of a run on, the reference frame is not handed over again -- its pyramids
            # are the ones the previous pair computed for it inside this timed region
            carried = carry[0] and j > 0
            if host_frames:
                upload(j + 2)  # the frame the NEXT feed needs: its upload overlaps this pair's nets
                ref = frame_on_device(j)
                p.enqueue_nets(j % SLOTS, None if carried else ref, frame_on_device(j + 1), d_feed)
                if PREFETCH:
                    p.prefetch_track(j % SLOTS)
                return
            if nets_mode:  # ping-pong A->B, B->A: every pair (and the rolled-over reference depth) is consistent
                p.enqueue_nets(j % SLOTS, None if carried else d_frames[j % 2], d_frames[1 - j % 2], d_feed)
                if PREFETCH:
                    p.prefetch_track(j % SLOTS)
                return
            p.enqueue_nets(j % SLOTS, d_frames[0], d_frames[1], d_feed)
            if PREFETCH:
                p.prefetch_track(j % SLOTS, d_sc[j % len(d_sc)][0], d_sc[j % len(d_sc)][1])

        # staged ramp (round 5): a run starts with RAMP pairs' nets in flight, not `ahead` -- with three started together the
        # first RandomState-ordered chain waits for the slowest of them (~10 ms to the first pose, 9 % of a 20-pair run);
        # with one, the first chain starts after one pair's latency and the pairs ahead are fed while it runs
        fed = 0
        while fed < min(RAMP, ahead, n):
            feed(fed)
            fed += 1
        for k in range(n):
            t_a = time.perf_counter()
            # the RandomState-ordered chain of pair k is enqueued first, the nets of the pairs up to k + ahead while it runs
            if nets_mode:
                p.track_begin(k % SLOTS)  # no overrides: keypoints / RANSAC / scale / PnP on the nets' own outputs
            else:
                f, dd, dp = d_sc[k % len(d_sc)]
                p.track_begin(k % SLOTS, f, dd, dp)  # E-tracker, or the PnP fallback when its pose is rejected
            while fed < n and fed <= k + ahead:
                feed(fed)
                fed += 1
            t_b = time.perf_counter()
            out = p.track_end(k % SLOTS)
            t_c = time.perf_counter()
            t_track.append(t_c)
            host_t[0] += t_b - t_a
            host_t[1] += t_c - t_b
            rel, _ = p.hybrid_pose(out, prev)
            prev = rel
            g = p.accumulate(g, rel)
            rel_all[k] = rel
            status[k] = out.status
        p.sync()
        return rel_all, status

    class PingPong:  # the sequence A, B, A, B, ... as an indexable of device frames
        def __getitem__(self, i):
            return d_frames[i % 2]

    seq_check = None
    if world == 1:
        run(pipe, args.warmup)
        device_sync()
        host_t[0] = host_t[1] = 0.0
        t0 = time.perf_counter()
        rel_all, status = run(pipe, args.steps)
        device_sync()
        dt = time.perf_counter() - t0
        if os.environ.get("DFVO_BENCH_TRACE"):
            sys.stderr.write("host ms/pair: enqueue_nets %.3f  track %.3f\n" % (host_t[0] * 1e3 / args.steps,
                                                                                host_t[1] * 1e3 / args.steps))
        gathered = dmod.allgather_poses(rel_all, status, 1, 0)
        n_total = args.steps
    else:
        # ONE sequence of world x steps pairs (ping-pong frames), contiguous chunk per rank from its 1-frame halo, per-pair
        # RandomState (df-vo_amd/sequence.py), ONE all-gather of pose + status rows, prefix composition on every rank
        n_total = world * args.steps
        frames = PingPong()
        smod.track_chunk(pipe, frames, 0, args.warmup, rng_mode="per_pair", carry_features=carry[0])
        dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        traj, gathered = smod.run_sequence(pipe, frames, n_total + 1, world, rank, dist, rng_mode="per_pair", carry_features=carry[0])
        device_sync()
        dist.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        status = gathered[:, 16].astype(np.int64)
        if rank == 0 and on_gpu:  # outside the timed region: the same sequence tracked by this rank alone
            rel1, st1 = smod.track_chunk(pipe, frames, 0, n_total, rng_mode="per_pair", carry_features=carry[0])
            same = bool(np.array_equal(rel1.reshape(n_total, 16)[st1 != 1], gathered[:, :16][st1 != 1]) and np.array_equal(st1, status))
            seq_check = {"pairs": int(n_total), "equal_to_single_rank_run": same,
                         "trajectory_end": [round(float(v), 4) for v in traj[-1][:3, 3]]}
            if not same:  # reported, not fatal: the line still carries the measured rate
                seq_check["max_abs_pose_diff"] = float(np.abs(rel1.reshape(n_total, 16) - gathered[:, :16]).max())
                sys.stderr.write("bench.py: the gathered poses of the %d-rank run differ from the single-rank run\n" % world)
    if (status == 2).any():
        raise SystemExit("bench.py: a pair needed the PnP fallback without a reference depth")
    net_flops = pipe.net_flops()  # issued by the last pair (carried mode: one Features pass)
    ahead = int(os.environ.get("DFVO_BENCH_AHEAD", "3"))
    steady = None
    if world == 1 and len(t_track) > ahead + 2:  # pairs after the first `ahead` (pipeline filled): the sustained rate
        steady = (len(t_track) - ahead) / (t_track[-1] - t_track[ahead - 1])
    net_flops_ref = net_flops  # the reference's work per pair: Features on both frames in every model call
    recomputed = None
    if rank == 0 and world == 1 and on_gpu and nets_mode and carry[0] and not args.no_other_legs:
        # the same workload with both frames of every pair through Features (third timed leg, same warm-up / steps / clock)
        carry[0] = False
        run(pipe, args.warmup)
        device_sync()
        tr = time.perf_counter()
        _, st_r = run(pipe, args.steps)
        device_sync()
        dtr = time.perf_counter() - tr
        net_flops_ref = pipe.net_flops()
        carry[0] = True
        recomputed = {"value": round(args.steps / dtr, 3), "unit": "frames/s", "ms_per_step": round(dtr / args.steps * 1e3, 3),
                      "algorithmic_gflop_per_pair": round(net_flops_ref / 1e9, 1),
                      "tracked_by_E": int((st_r == 0).sum()), "tracked_by_PnP": int((st_r == 3).sum())}

    roof = None
    if rank == 0 and world == 1 and not args.no_roofline:
        # per-launch durations of the conv kernel family, HIP events on the launch streams, graphs off
        # (one pair in flight at a time here: overlapping passes would stretch each other's launches)
        pipe.set_graph(0)
        pipe.enqueue_nets(0, d_frames[0], d_frames[1], d_feed)
        pipe.sync()
        lib = capi.lib()
        capi.check(lib.dfvo_conv_profile_begin())
        nprof = 3
        for _ in range(nprof):
            pipe.enqueue_nets(0, d_frames[0], d_frames[1], d_feed)
            pipe.sync()
        ms = np.zeros(24)
        fl = np.zeros(24)
        ln = np.zeros(24, np.int32)
        by = np.zeros(24)
        capi.check(lib.dfvo_conv_profile_end_bytes(capi.as_ptr(ms), capi.as_ptr(fl), capi.as_ptr(ln), capi.as_ptr(by)))
        pipe.set_graph(1)
        dom = int(np.argmax(ms))
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12
        fam = fl.sum() / (ms.sum() * 1e-3) / 1e12
        # exact fp32: the fp32-MFMA peak.  f16x3: three f16 products per fp32 product, so the ceiling for USEFUL
        # fp32-equivalent FLOPs is the dense f16 peak divided by three
        terms = {"fp32": 0, "f16x3": 3, "f16": 1}[args.conv_precision]
        if terms and dom < 19:
            terms = 0  # a kernel of the exact fp32 family dominates although the window layers run split: price it as fp32
        peak = PEAK_F32_MFMA_TFLOPS if not terms else PEAK_F16_MFMA_TFLOPS / terms
        roof = {"bound": "mfma", "kernel": CFG_NAMES[dom],
                "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "avg_launch_us": round(ms[dom] * 1e3 / max(1, ln[dom]), 2), "launches_per_pair": int(ln[dom] // nprof),
                "share_of_conv_time": round(float(ms[dom] / ms.sum()), 3),
                "conv_family_achieved": round(fam, 2), "conv_family_ms_per_pair": round(float(ms.sum() / nprof), 3),
                "algorithmic_gflop_per_pair": round(net_flops / 1e9, 1),
                "by_config": [{"kernel": CFG_NAMES[i], "ms_per_pair": round(float(ms[i] / nprof), 3),
                               "launches_per_pair": int(ln[i] // nprof), "gflop_per_pair": round(float(fl[i] / nprof / 1e9), 1),
                               "tflops": round(float(fl[i] / (ms[i] * 1e-3) / 1e12), 1)}
                              for i in np.argsort(-ms) if ln[i] > 0]}
        # SURVEY 8d bounds the streaming layers, the correlation, the warps and the resizes by HBM: the conv family's streaming
        # configuration (1x1 / stride 2 / tap-window layers) priced live -- algorithmic bytes (input map + output map + weights,
        # each once, from the launches' shapes) over the HIP-event durations of the same launches -- and the non-conv streaming
        # kernels from the committed rocprofv3 passes (bytes from the PMC pass, duration from the kernel-trace statistics)
        if ln[20] > 0:
            roof["hbm"] = {"bound": "hbm", "kernel": CFG_NAMES[20], "achieved": round(float(by[20] / (ms[20] * 1e-3) / 1e9), 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(float(by[20] / (ms[20] * 1e-3) / 1e9 / PEAK_HBM_GBS), 4),
                           "traffic": None, "avg_launch_us": round(float(ms[20] * 1e3 / ln[20]), 2),
                           "launches_per_pair": int(ln[20] // nprof),
                           "algorithmic_mb_per_launch": round(float(by[20] / ln[20] / 1e6), 2),
                           "note": "achieved = algorithmic bytes / HIP-event duration, live; traffic and the non-conv streaming "
                                   "kernels (other_kernels) from the committed rocprofv3 passes"}
            roof["hbm"]["other_kernels"] = committed_streaming_kernels()
        # HBM-side bytes per launch of that kernel: hardware counters cannot be read from inside the process, so they come
        # from the committed rocprofv3 --pmc passes over this same command (tools/profile.sh -> tools/pmc_traffic.py:
        # FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes); None when no committed profile holds the kernel.  The
        # profile records the library build it was taken on: a stale file is reported as such, not silently used.
        try:
            prof_all = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            prof = prof_all["kernels"]
            key = {19: "conv_win_f16s", 20: "conv_gemm_f16s_kernel<", 21: "conv_gemm_f16s_kernel<"}.get(dom) or CFG_NAMES[dom].split(" ")[0].replace(
                "conv_igemm_f32<", "conv_igemm_f32_kernel<").replace("conv_win3_f32<", "conv_win_f32_kernel<").replace(
                "conv_win_f32<", "conv_win_f32_kernel<")
            cand = [k for k in prof if k.replace(" ", "").startswith(key.rstrip(">").replace(" ", ""))]
            if roof.get("hbm"):
                cs = [k for k in prof if k.replace(" ", "").startswith("conv_taps_f16s_kernel") or
                      k.replace(" ", "").startswith("conv_gemm_f16s_kernel<4,1,")]
                if cs:
                    nds = sum(prof[k]["dispatches"] for k in cs)
                    roof["hbm"]["traffic"] = round(sum(prof[k]["hbm_bytes_per_dispatch"] * prof[k]["dispatches"] for k in cs) / nds)
            if cand:
                nd = sum(prof[k]["dispatches"] for k in cand)
                roof["traffic"] = round(sum(prof[k]["hbm_bytes_per_dispatch"] * prof[k]["dispatches"] for k in cand) / nd)
                roof["traffic_unit"] = "bytes per launch, mean over %d profiled launches (rocprofv3 PMC, profiles/%s)" % (nd, PMC_FILE)
                src_now = kernel_source_digest()
                if prof_all.get("kernel_source_digest") not in (None, src_now):
                    roof["traffic_stale"] = "profile taken on kernel sources %s, this build is %s" % (prof_all.get("kernel_source_digest"), src_now)
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass

    exact = None
    if rank == 0 and world == 1 and not args.no_exact_leg and args.conv_precision != "fp32" and nets_mode and on_gpu:
        # the same workload with every layer on the exact fp32-MFMA kernels: a second pipeline (the precision is fixed when
        # the layers are packed), same warm-up and step count, timed the same way
        pipe.close()
        pipe = None
        capi.check(capi.lib().dfvo_set_conv_precision(b"fp32"))
        pipe_x = pmod.TrackingPipeline(H, W, 192, 640, K, fsd, dsd, **popts)
        run(pipe_x, args.warmup)
        device_sync()
        tx = time.perf_counter()
        _, st_x = run(pipe_x, args.steps)
        device_sync()
        dtx = time.perf_counter() - tx
        exact = {"value": round(args.steps / dtx, 3), "unit": "frames/s", "ms_per_step": round(dtx / args.steps * 1e3, 3),
                 "dtype": "f32 (every layer on v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate)",
                 "tracked_by_E": int((st_x == 0).sum()), "tracked_by_PnP": int((st_x == 3).sum())}
        pipe_x.close()
        capi.check(capi.lib().dfvo_set_conv_precision(args.conv_precision.encode()))
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline_nets(seq["frames"], fsd, dsd, K, H, W, n_pairs=args.cpu_pairs) if nets_mode else cpu_baseline(syn, H, W, scenes)
    seen = ranks_seen(dist, world, rank, local_rank, torch)
    if pipe is not None:
        pipe.close()
    if rank == 0:
        n_e = int((status == 0).sum())
        if roof is not None:
            roof["whole_pair_tflops"] = round(net_flops * n_total / dt / 1e12, 2)  # issued net FLOPs / timed wall time
            roof["whole_pair_tflops_two_features_passes"] = round(net_flops_ref * n_total / dt / 1e12, 2)
            roof["algorithmic_gflop_per_pair"] = round(net_flops / 1e9, 1)
            roof["algorithmic_gflop_per_pair_two_features_passes"] = round(net_flops_ref / 1e9, 1)  # both frames through Features once
            # the reference's model call runs Features on FOUR images per pair (lite_flow_net.py:285-325 with batch 2 on both
            # arguments: forward and backward samples each carry both frames): two more single-frame passes than the figure above
            roof["algorithmic_gflop_per_pair_reference"] = (round((net_flops_ref + 2 * (net_flops_ref - net_flops)) / 1e9, 1)
                                                            if recomputed is not None else None)
            roof["note"] = ("achieved/frac: the conv tile configuration with the largest time share, per-launch HIP-event "
                            "durations, graphs off, ONE pair in flight (conv_family_ms_per_pair sums those and exceeds "
                            "ms_per_step, whose timed region overlaps two flow-net instances and the solver stage); "
                            "whole_pair_tflops = issued net FLOPs / timed wall time; traffic: rocprofv3 PMC pass over this "
                            "command, committed under profiles/ (counters cannot be read from inside the process)")
        net_h, net_w = syn._net_size(H, W)
        if nets_mode:
            si = ("the nets' own outputs (product data path, no overrides): coded tunnel-world frames, '%s' encoding, "
                  "tracked ping-pong A->B, B->A" % code_mode)
        else:
            si = ("synthetic rigid-scene flow/consistency/depth overrides (random-weight nets give incoherent flow); net "
                  "outputs are computed in the timed region")
        line = {
            "metric": "KITTI-odom frames/sec (DF-VO per-pair tracking hot path: monodepth2 + LiteFlowNet fwd/bwd + "
                      "kp selection + E/H RANSAC + scale)",
            "value": round(n_total / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16x3": "f32 (f16x3 split products: two f16 planes per operand = 22 mantissa bits, three exact "
                                                "products per term, fp32 accumulate; direct one- / two-channel heads exact fp32)",
                      "f16": "f16 (operands rounded to f16, ONE product per term on v_mfma_f32_32x32x16_f16, fp32 accumulate and fp32 "
                             "activations between layers; direct one- / two-channel heads exact fp32) -- not fp32-class"
                      }[args.conv_precision],
            "data": "synthetic",
            "config": {"workload": "%dx%d frame pairs%s (flow net %dx%d batch 2; device LANCZOS resize + depth net 192x640), "
                                   "local_bestN %d keypoints, findHomography + 5x findEssentialMat(%d-iteration budget) + GRIC + "
                                   "recoverPose + depth-ratio scale RANSAC, PnP fallback (5x solvePnPRansac) where the "
                                   "reference takes it" % (W, H, " (KITTI seq-09 size)" if (H, W) == (376, 1241) else "", net_h,
                                                           net_w, args.kp_bestn, args.e_max_iters),
                       "conv_precision": args.conv_precision, "frames_per_gpu": args.steps,
                       "feature_pyramids": ("carried: every new frame runs through the flow net's Features once, inside the timed "
                                            "region; the next pair reads them as its reference frame's (identical values)"
                                            if carry[0] else "recomputed: both frames of every pair run through Features"),
                       "frames": "pinned host memory, uploaded on a copy stream inside the timed region" if host_frames else "resident in HBM",
                       "parallelism": ("one sequence of %d pairs, contiguous chunk + 1-frame halo per rank, per-pair RandomState, one "
                                       "all-gather of poses, prefix composition" % n_total) if world > 1 else
                                      "single rank, sequential RandomState (the reference's mode)",
                       "solver_inputs": si,
                       "tracked_by_E": n_e, "tracked_by_PnP": int((status == 3).sum()),
                       "constant_motion": int((status == 1).sum()), "gathered_poses": int(gathered.shape[0]),
                       "ranks_seen": len(set(r[0] for r in seen)), "devices_seen": len(set((r[1], r[2], r[3]) for r in seen))},
            "steady_state": None if steady is None else {"value": round(steady, 3), "unit": "frames/s",
                                                         "note": "pairs after the first %d (nets running ahead of the solver "
                                                                 "stage), host clock at the return of track()" % ahead},
            "exact_fp32": exact, "features_recomputed": recomputed, "sequence_check": seq_check,
            "frames_host": frames_host, "dropin_surface": dropin, "other_configs": others,
            "roofline": roof, "cpu_baseline": base}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
