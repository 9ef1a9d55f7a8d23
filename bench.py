#!/usr/bin/env python
"""bench.py -- KITTI-sized frame pairs per second through the DF-VO tracking hot path on MI355X.

One step = one frame pair (1241x376): monodepth2 depth + LiteFlowNet forward/backward flow + consistency
(HIP fp32-MFMA nets), local_bestN keypoint selection, homography + 5 x five-point RANSAC + GRIC +
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
directly, bench.py re-launches itself under torch.distributed.run with N ranks on 127.0.0.1.

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
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix (f32 in / f32 acc)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table: BF16 MFMA dense (only used for the opt-in split-precision modes)
CFG_NAMES = ["conv_igemm_f32<2,2,4,4> 128x128", "conv_igemm_f32<1,4,2,2> 32x128", "conv_igemm_f32<4,1,4,4> 256x64",
             "conv_igemm_f32<2,2,2,2> 64x64", "conv_igemm_f32<4,1,4,2> 256x32", "conv_igemm_f32<2,2,2,1> 64x32",
             "conv_igemm_f32<4,1,4,1> 256x16", "conv_igemm_f32<4,1,1,1> 64x16", "conv_igemm_f32<1,4,4,2> 64x128",
             "conv_igemm_f32<2,2,4,2> 128x64", "conv_igemm_f32<4,1,2,2> 128x32", "conv_igemm_f32<4,1,2,1> 128x16",
             "conv_win3_f32<2,2,4,4> (8x16)x128", "conv_win3_f32<2,2,4,2> (8x16)x64", "conv_win3_f32<4,1,2,2> (8x16)x32",
             "conv_win3_f32<1,4,4,2> (4x16)x128", "conv_head_f32<7> 7x7 heads (direct)",
             "conv_head_f32<5> 5x5 heads (direct)", "conv_head_f32<3> 3x3 heads (direct)",
             "conv_win_f16s (4|8 x 32) x 128|64|32 (f16 hi/lo planes, v_mfma_f32_32x32x16_f16)",
             "conv_gemm_f16s streaming (1x1, k x 1, stride 2, 7x7 layers; 32 px x 32|64 couts per wave, f16 hi/lo planes)",
             "conv_gemm_f16s K-sliced (small maps: pyramid levels 5-6, depth net inner layers; in-workgroup ordered reduction)",
             "(unused)", "(unused)"]


def cpu_baseline_nets(frames, fsd, dsd, K, H, W, n_pairs=3):
    """the oracle's image-level frame loop (oracle/pipeline_np.py: torch-CPU nets + C/numpy solvers) over the same coded
    frames, ping-pong like the timed region, on the host cores"""
    import torch
    from oracle import pipeline_np as P
    cores = torch.get_num_threads()
    seq = [frames[i % 2] for i in range(n_pairs + 1)]
    t0 = time.time()
    r = P.track_sequence(seq, fsd, dsd, K, seed=4869)
    dt = time.time() - t0
    return {"value": n_pairs / dt, "unit": "frames/s", "cores": int(cores), "kind": "port",
            "sample": "%d frame pairs %dx%d through the oracle frame loop (torch-CPU fp32 monodepth2 + LiteFlowNet fwd+bwd, "
                      "C/numpy solvers on their outputs; tracked %s), %.1f s" % (n_pairs, W, H, "/".join(r["status"]), dt)}


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


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-precision", default=os.environ.get("DFVO_CONV_PRECISION", "f16x3"), choices=["fp32", "f16x3", "bf16x6", "bf16x3"],
                    help="arithmetic of the 3x3 window layers: f16x3 (default: fp32-class split kernel, the whole -m gpu net / "
                         "pipeline parity suite runs green on it at the exact path's tolerances), fp32 (exact fp32 MFMA), "
                         "bf16x6 / bf16x3 (round-1 experiments) -- reported in dtype and config.conv_precision")
    ap.add_argument("--solver-inputs", default="nets", choices=["nets", "synthetic"],
                    help="nets: the solver stage consumes the nets' own outputs (coded-world frames, the product data path); "
                         "synthetic: random-weight nets + a synthetic rigid-scene flow/consistency/depth triple (round-1 mode)")
    ap.add_argument("--e-max-iters", type=int, default=1000, help="findEssentialMat hypothesis budget (config 5: 8192)")
    ap.add_argument("--kp-bestn", type=int, default=2000, help="kp_selection.local_bestN.num_bestN (config 5: 20000)")
    args = ap.parse_args(argv)
    os.environ["DFVO_CONV_PRECISION"] = args.conv_precision  # read once by the library when the layers are packed
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    rank, world, local_rank, dist = world_setup(args, torch)
    on_gpu = dist is None or dist.get_backend() == "nccl"  # (gloo: CPU test of the multi-rank logic, tests/test_bench_cpu.py)
    if world == 1:
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
    dev = to_device
    d_feed = None  # the pipeline resizes the current frame itself (device LANCZOS, bit-exact with Pillow)
    popts = dict(seed=4869 ^ rank, e_max_iters=args.e_max_iters, kp_num_bestN=args.kp_bestn)  # per-rank RandomState: DP mode
    # Everything the host prepares (frames, scenes, weights) first, THEN the pipeline, THEN the device copies of the inputs:
    # the pair rate depends on the order in which the process creates its HIP streams (133 vs 109 frames/s, exact fp32, same
    # binary and box: which compute pipes the pipeline's streams share follows their creation order, see
    # TrackerBuffers::init and DESIGN.md section 5)
    if nets_mode:
        # coded tunnel world: multiplexed encoding when the frame needs no input resize, potential encoding otherwise
        code_mode = "mux" if syn._net_size(H, W) == (H, W) else "pot"
        # (potential encoding: the sideways drive, for which it yields E-tracked pairs -- tunnel_poses_lateral)
        seq = syn.coded_tunnel_sequence(H, W, 2, mode=code_mode, step=1.0, seed=7 + rank,
                                        poses=None if code_mode == "mux" else syn.tunnel_poses_lateral(2, 0.4))
        K = seq["K"]
        fsd, dsd = syn.crafted_liteflownet_state_dict(H, W, code_mode), syn.crafted_monodepth2_state_dict()
        scenes = None
        h_frames = [seq["frames"][0], seq["frames"][1]]
    else:
        scenes = [syn.rigid_scene(H, W, seed=100 + 7 * rank + i) for i in range(4)]
        K = scenes[0]["K"]
        fsd, dsd = syn.liteflownet_state_dict(4869), syn.monodepth2_state_dict(4869)
        h_frames = list(syn.image_pair(H, W, seed=1 + rank))
    pipe = pmod.TrackingPipeline(H, W, 192, 640, K, fsd, dsd, **popts)
    d_frames = [dev(f) for f in h_frames]
    if not nets_mode:
        d_sc = [(dev(s["flow"]), dev(s["diff"]), dev(s["depth_cur"])) for s in scenes]
        d_ref_depth = dev(scenes[0]["depth_ref"])

    host_t = [0.0, 0.0]  # host seconds inside enqueue_nets / track (DFVO_BENCH_TRACE=1 prints them)

    def run(n):
        """software pipeline: the nets of pairs k+1, k+2 are enqueued before the solver stage of pair k blocks the host"""
        g = np.eye(4)
        prev = np.eye(4)
        rel_all = np.zeros((n, 4, 4))
        status = np.zeros(n, np.int64)
        if n == 0:
            return rel_all, status
        if nets_mode:
            pipe.set_ref_image(d_frames[0])  # depth of the first reference frame (PnP fallback input), from the frame itself
        else:
            pipe.set_ref_depth(depth=d_ref_depth)
        # nets run `ahead` pairs ahead of the solver stage (<= SLOTS - 1): with 3, each of the two flow-net instances always
        # has its next pair queued behind the current one while track(k) blocks the host
        ahead = int(os.environ.get("DFVO_BENCH_AHEAD", "3"))
        def feed(j):  # nets of pair j, then the RNG-independent half of its solver stage right behind them
            if nets_mode:  # ping-pong A->B, B->A: every pair (and the rolled-over reference depth) is consistent
                pipe.enqueue_nets(j % SLOTS, d_frames[j % 2], d_frames[1 - j % 2], d_feed)
                if PREFETCH:
                    pipe.prefetch_track(j % SLOTS)
                return
            pipe.enqueue_nets(j % SLOTS, d_frames[0], d_frames[1], d_feed)
            if PREFETCH:
                pipe.prefetch_track(j % SLOTS, d_sc[j % len(d_sc)][0], d_sc[j % len(d_sc)][1])

        for j in range(min(ahead, n)):
            feed(j)
        for k in range(n):
            t_a = time.perf_counter()
            if k + ahead < n:
                feed(k + ahead)
            t_b = time.perf_counter()
            if nets_mode:
                out = pipe.track(k % SLOTS)  # no overrides: keypoints / RANSAC / scale / PnP on the nets' own outputs
            else:
                f, dd, dp = d_sc[k % len(d_sc)]
                out = pipe.track(k % SLOTS, f, dd, dp)  # E-tracker, or the PnP fallback when its pose is rejected
            host_t[0] += t_b - t_a
            host_t[1] += time.perf_counter() - t_b
            rel, _ = pipe.hybrid_pose(out, prev)
            prev = rel
            g = pipe.accumulate(g, rel)
            rel_all[k] = rel
            status[k] = out.status
        pipe.sync()
        return rel_all, status

    run(args.warmup)
    if dist is not None:
        dist.barrier()
    device_sync()
    host_t[0] = host_t[1] = 0.0
    t0 = time.perf_counter()
    rel_all, status = run(args.steps)
    if os.environ.get("DFVO_BENCH_TRACE"):
        sys.stderr.write("host ms/pair: enqueue_nets %.3f  track %.3f\n" % (host_t[0] * 1e3 / args.steps,
                                                                            host_t[1] * 1e3 / args.steps))
    gathered = dmod.allgather_poses(rel_all, status, world, rank, dist, "cuda" if on_gpu else "cpu")  # one RCCL all-gather of the chunk's poses
    if (status == 2).any():
        raise SystemExit("bench.py: a pair needed the PnP fallback without a reference depth")
    device_sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    net_flops = pipe.net_flops()

    roof = None
    if rank == 0 and not args.no_roofline:
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
        capi.check(lib.dfvo_conv_profile_end(capi.as_ptr(ms), capi.as_ptr(fl), capi.as_ptr(ln)))
        pipe.set_graph(1)
        dom = int(np.argmax(ms))
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12
        fam = fl.sum() / (ms.sum() * 1e-3) / 1e12
        # exact fp32: the fp32-MFMA peak.  Opt-in split modes: the window layers issue 4 (bf16x3) / 6 (bf16x6) bf16 products
        # per fp32 product, so the ceiling for USEFUL fp32-equivalent FLOPs is the dense bf16 peak divided by that
        terms = {"fp32": 0, "f16x3": 3, "bf16x3": 4, "bf16x6": 6}[args.conv_precision]
        if terms and dom < 19 and args.conv_precision == "f16x3":
            terms = 0  # a kernel of the exact fp32 family dominates although the window layers run split: price it as fp32
        peak = PEAK_F32_MFMA_TFLOPS if not terms else PEAK_BF16_MFMA_TFLOPS / terms
        roof = {"bound": "mfma", "kernel": CFG_NAMES[dom].replace("conv_win3_f32", "conv_win3_f32" if not terms else "conv_win3_bf16s"),
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
    if roof is not None:
        # HBM-side bytes per launch of that kernel: hardware counters cannot be read from inside the process, so they come
        # from the committed rocprofv3 --pmc passes over this same command (tools/r2c_profile.sh -> tools/pmc_traffic.py:
        # FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes); None when no committed profile holds the kernel
        try:
            pmc_file = "r2c_pmc_bench.json"
            prof = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["kernels"]
            key = {19: "conv_win_f16s_kernel<", 20: "conv_gemm_f16s_kernel<", 21: "conv_gemm_f16s_kernel<"}.get(dom) or CFG_NAMES[dom].split(" ")[0].replace(
                "conv_igemm_f32<", "conv_igemm_f32_kernel<").replace("conv_win3_f32<", "conv_win_f32_kernel<").replace(
                "conv_win_f32<", "conv_win_f32_kernel<")
            cand = [k for k in prof if k.replace(" ", "").startswith(key.rstrip(">").replace(" ", ""))]
            if cand:
                nd = sum(prof[k]["dispatches"] for k in cand)
                roof["traffic"] = round(sum(prof[k]["hbm_bytes_per_dispatch"] * prof[k]["dispatches"] for k in cand) / nd)
                roof["traffic_unit"] = "bytes per launch, mean over %d profiled launches (rocprofv3 PMC, profiles/%s)" % (nd, pmc_file)
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline_nets(seq["frames"], fsd, dsd, K, H, W) if nets_mode else cpu_baseline(syn, H, W, scenes)
    seen = ranks_seen(dist, world, rank, local_rank, torch)
    pipe.close()

    if rank == 0:
        n_e = int((status == 0).sum())
        if roof is not None:
            roof["whole_pair_tflops"] = round(net_flops * world * args.steps / dt / 1e12, 2)  # issued net FLOPs / timed wall time
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
            "value": round(world * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16x3": "f32 (f16x3 split products in the 3x3 window layers: two f16 planes per operand = 22 "
                                                "mantissa bits, three exact products per term, fp32 accumulate; all other layers exact fp32 MFMA)"
                      }.get(args.conv_precision, "f32 accumulate, %s split-bf16 products in the 3x3 window layers (opt-in)" % args.conv_precision),
            "data": "synthetic",
            "config": {"workload": "%dx%d frame pairs%s (flow net %dx%d batch 2; device LANCZOS resize + depth net 192x640), "
                                   "local_bestN %d keypoints, findHomography + 5x findEssentialMat(%d-iteration budget) + GRIC + "
                                   "recoverPose + depth-ratio scale RANSAC, PnP fallback (5x solvePnPRansac) where the "
                                   "reference takes it" % (W, H, " (KITTI seq-09 size)" if (H, W) == (376, 1241) else "", net_h,
                                                           net_w, args.kp_bestn, args.e_max_iters),
                       "conv_precision": args.conv_precision, "frames_per_gpu": args.steps,
                       "parallelism": "frame-batch DP x%d, one all-gather of poses" % world,
                       "solver_inputs": si,
                       "tracked_by_E": n_e, "tracked_by_PnP": int((status == 3).sum()),
                       "constant_motion": int((status == 1).sum()), "gathered_poses": int(gathered.shape[0]),
                       "ranks_seen": len(set(r[0] for r in seen)), "devices_seen": len(set((r[1], r[2], r[3]) for r in seen))},
            "roofline": roof, "cpu_baseline": base}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
