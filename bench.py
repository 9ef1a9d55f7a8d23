#!/usr/bin/env python
"""bench.py -- KITTI-sized frame pairs per second through the DF-VO tracking hot path on MI355X.

One step = one frame pair (1241x376): monodepth2 depth + LiteFlowNet forward/backward flow + consistency
(HIP fp32-MFMA nets), local_bestN keypoint selection, homography + 5 x five-point RANSAC + GRIC +
recoverPose, depth-ratio scale RANSAC (PnP fallback where the reference takes it), pose out.  Inputs (the two uint8
frames) are resident in HBM before the timed region; the Pillow-exact LANCZOS resize of the depth input runs on the
device inside it.  Random-weight nets give incoherent flow, so the solver stage is
fed a synthetic rigid-scene flow / consistency / depth triple of the same shape (also HBM resident) and
does the full work it does on KITTI; the nets' own outputs are still computed inside the timed region.

    python bench.py [--gpus N --steps K --warmup W]      (N > 1: launched by torch.distributed.run)

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
             "conv_head_f32<5> 5x5 heads (direct)", "conv_head_f32<3> 3x3 heads (direct)"]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=376)
    ap.add_argument("--width", type=int, default=1241)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--conv-precision", default=os.environ.get("DFVO_CONV_PRECISION", "fp32"), choices=["fp32", "bf16x6", "bf16x3"],
                    help="fp32 (default, the parity-gated exact path); bf16x6 / bf16x3: opt-in split-precision MFMA modes "
                         "of the 3x3 window layers (DESIGN.md section 3) -- reported in config.conv_precision, never the default")
    args = ap.parse_args()
    os.environ["DFVO_CONV_PRECISION"] = args.conv_precision  # read once by the library when the layers are packed

    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)

    pkg = importlib.import_module("df-vo_amd")  # noqa: F841
    capi = importlib.import_module("df-vo_amd.capi")
    syn = importlib.import_module("df-vo_amd.synthetic")
    pmod = importlib.import_module("df-vo_amd.pipeline")
    dmod = importlib.import_module("df-vo_amd.dist")
    capi.check(capi.lib().dfvo_set_device(local_rank if world > 1 else 0))

    H, W = args.height, args.width
    scenes = [syn.rigid_scene(H, W, seed=100 + 7 * rank + i) for i in range(4)]
    K = scenes[0]["K"]
    pipe = pmod.TrackingPipeline(H, W, 192, 640, K, syn.liteflownet_state_dict(4869), syn.monodepth2_state_dict(4869),
                                 seed=4869 ^ rank)  # per-rank stream: "per-frame-seed" mode of the DP driver
    ref, cur = syn.image_pair(H, W, seed=1 + rank)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_ref, d_cur = dev(ref), dev(cur)
    d_feed = None  # the pipeline resizes the current frame itself (device LANCZOS, bit-exact with Pillow)
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
        pipe.set_ref_depth(depth=d_ref_depth)  # depth of the first reference frame (PnP fallback input)
        # nets run `ahead` pairs ahead of the solver stage (<= SLOTS - 1): with 3, each of the two flow-net instances always
        # has its next pair queued behind the current one while track(k) blocks the host
        ahead = int(os.environ.get("DFVO_BENCH_AHEAD", "3"))
        def feed(j):  # nets of pair j, then the RNG-independent half of its solver stage right behind them
            pipe.enqueue_nets(j % SLOTS, d_ref, d_cur, d_feed)
            if PREFETCH:
                pipe.prefetch_track(j % SLOTS, d_sc[j % len(d_sc)][0], d_sc[j % len(d_sc)][1])

        for j in range(min(ahead, n)):
            feed(j)
        for k in range(n):
            t_a = time.perf_counter()
            if k + ahead < n:
                feed(k + ahead)
            t_b = time.perf_counter()
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
    torch.cuda.synchronize()
    host_t[0] = host_t[1] = 0.0
    t0 = time.perf_counter()
    rel_all, status = run(args.steps)
    if os.environ.get("DFVO_BENCH_TRACE"):
        sys.stderr.write("host ms/pair: enqueue_nets %.3f  track %.3f\n" % (host_t[0] * 1e3 / args.steps,
                                                                            host_t[1] * 1e3 / args.steps))
    gathered = dmod.allgather_poses(rel_all, status, world, rank, dist)  # one RCCL all-gather of the chunk's poses
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    net_flops = pipe.net_flops()

    roof = None
    if rank == 0 and not args.no_roofline:
        # per-launch durations of the conv kernel family, HIP events on the launch streams, graphs off
        # (one pair in flight at a time here: overlapping passes would stretch each other's launches)
        pipe.set_graph(0)
        pipe.enqueue_nets(0, d_ref, d_cur, d_feed)
        pipe.sync()
        lib = capi.lib()
        capi.check(lib.dfvo_conv_profile_begin())
        nprof = 3
        for _ in range(nprof):
            pipe.enqueue_nets(0, d_ref, d_cur, d_feed)
            pipe.sync()
        ms = np.zeros(19)
        fl = np.zeros(19)
        ln = np.zeros(19, np.int32)
        capi.check(lib.dfvo_conv_profile_end(capi.as_ptr(ms), capi.as_ptr(fl), capi.as_ptr(ln)))
        pipe.set_graph(1)
        dom = int(np.argmax(ms))
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12
        fam = fl.sum() / (ms.sum() * 1e-3) / 1e12
        # exact fp32: the fp32-MFMA peak.  Opt-in split modes: the window layers issue 4 (bf16x3) / 6 (bf16x6) bf16 products
        # per fp32 product, so the ceiling for USEFUL fp32-equivalent FLOPs is the dense bf16 peak divided by that
        terms = {"fp32": 0, "bf16x3": 4, "bf16x6": 6}[args.conv_precision]
        peak = PEAK_F32_MFMA_TFLOPS if not terms else PEAK_BF16_MFMA_TFLOPS / terms
        roof = {"bound": "mfma", "kernel": CFG_NAMES[dom].replace("conv_win3_f32", "conv_win3_f32" if not terms else "conv_win3_bf16s"),
                "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "avg_launch_us": round(ms[dom] * 1e3 / max(1, ln[dom]), 2), "launches_per_pair": int(ln[dom] // nprof),
                "share_of_conv_time": round(float(ms[dom] / ms.sum()), 3),
                "conv_family_achieved": round(fam, 2), "conv_family_ms_per_pair": round(float(ms.sum() / nprof), 3),
                "algorithmic_gflop_per_pair": round(net_flops / 1e9, 1)}
    if roof is not None:
        # HBM-side bytes per launch of that kernel from the committed rocprofv3 --pmc passes over this same command
        # (tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE); None when no profile matches the kernel
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r1z_pmc_bench.json")))["kernels"]
            key = CFG_NAMES[dom].split(" ")[0].replace("conv_igemm_f32<", "conv_igemm_f32_kernel<").replace(
                "conv_win3_f32<", "conv_win_f32_kernel<").replace("conv_win_f32<", "conv_win_f32_kernel<")
            cand = [k for k in prof if k.replace(" ", "").startswith(key.rstrip(">").replace(" ", ""))]
            if cand:
                roof["traffic"] = round(prof[cand[0]]["hbm_bytes_per_dispatch"])
                roof["traffic_unit"] = "bytes per launch (rocprofv3 PMC, profiles/r1z_pmc_bench.json)"
        except (OSError, KeyError, ValueError):
            pass
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline(syn, H, W, scenes)
    pipe.close()

    if rank == 0:
        n_e = int((status == 0).sum())
        line = {
            "metric": "KITTI-odom frames/sec (DF-VO per-pair tracking hot path: monodepth2 + LiteFlowNet fwd/bwd + "
                      "kp selection + E/H RANSAC + scale)",
            "value": round(world * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_precision == "fp32" else "f32 accumulate, %s split-bf16 products in the 3x3 window layers (opt-in)" % args.conv_precision,
            "data": "synthetic",
            "config": {"workload": "KITTI seq-09-sized 1241x376 frame pairs (flow net 384x1248 batch 2; device LANCZOS "
                                   "resize + depth net 192x640), 2000 keypoints, findHomography + 5x findEssentialMat"
                                   "(1000-iteration budget) + GRIC + recoverPose + depth-ratio scale RANSAC, PnP "
                                   "fallback (5x solvePnPRansac) where the reference takes it",
                       "conv_precision": args.conv_precision, "frames_per_gpu": args.steps, "parallelism": "frame-batch DP x%d, one all-gather of poses" % world,
                       "solver_inputs": "synthetic rigid-scene flow/consistency/depth (random-weight nets give "
                                        "incoherent flow); net outputs are computed in the timed region",
                       "tracked_by_E": n_e, "tracked_by_PnP": int((status == 3).sum()), "gathered_poses": int(gathered.shape[0])},
            "roofline": roof, "cpu_baseline": base}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
