"""CPU (gloo, world_size 2): the multi-rank logic of bench.py -- world setup from the torchrun environment, ONE sequence
of world x steps pairs tracked as contiguous chunks from each rank's halo frame (df-vo_amd/sequence.py run_sequence), the
single pose all-gather, max-over-ranks timing, rank 0's JSON line (n_gpus, ranks_seen) -- with the device pipeline replaced
by a stub (no GPU here), and the `--gpus N` self-launch command line."""
import importlib
import io
import json
import os
import socket
import sys
from contextlib import redirect_stdout

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Out:
    def __init__(self, k, seed):
        self.status, self.scale, self.n_kp, self.good_kp_found = 0, 1.0 + 0.01 * k, 2000, 1
        self.R = [1, 0, 0, 0, 1, 0, 0, 0, 1]
        self.t = [0.0, 0.0, float(seed % 7 + 1)]


class StubPipeline:
    """the surface bench.py drives (df-vo_amd/pipeline.py TrackingPipeline), answered on the host"""
    calls = []

    def __init__(self, H, W, fh, fw, K, fsd, dsd, **opts):
        self.seed_ = opts.get("seed", 0)
        self.k = 0

    def set_ref_image(self, img):
        StubPipeline.calls.append("ref")

    def seed(self, seed):  # per-pair RandomState of the data-parallel mode (df-vo_amd/sequence.py)
        self.seed_ = seed

    def enqueue_nets(self, slot, ref, cur, feed=None):
        assert 0 <= slot < 4

    def prefetch_track(self, slot, *a):
        pass

    def track(self, slot, *a):
        self.k += 1
        return _Out(self.k, self.seed_)

    def track_begin(self, slot, *a):
        self.k += 1

    def track_end(self, slot):
        return _Out(self.k, self.seed_)

    def sync(self):
        pass

    def net_flops(self):
        return 1e9

    def close(self):
        pass


def _worker(rank, world, port, q, extra=()):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DFVO_BENCH_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    pmod = importlib.import_module("df-vo_amd.pipeline")
    StubPipeline.hybrid_pose = staticmethod(pmod.TrackingPipeline.hybrid_pose)  # (host-side helpers of the real class)
    StubPipeline.accumulate = staticmethod(pmod.TrackingPipeline.accumulate)
    pmod.TrackingPipeline = StubPipeline
    bench.to_device = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    bench.device_sync = lambda: None
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "5", "--warmup", "1", "--height", "64", "--width", "96",
                    "--no-roofline", "--no-cpu-baseline"] + list(extra))
    q.put((rank, buf.getvalue()))


def test_bench_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[1].strip() == ""  # only rank 0 prints
    line = json.loads(outs[0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["ranks_seen"] == 2 and line["config"]["gathered_poses"] == 10  # every rank composed all 10 pairs
    assert line["unit"] == "frames/s" and line["value"] > 0
    assert abs(line["value"] - 2 * 5 / (line["ms_per_step"] * 5e-3)) < 1e-2 * line["value"]  # whole-job aggregate


def test_bench_sequences_mode_two_ranks_gloo():
    """bench.py --sequences kitti-lengths (BASELINE config 3) under gloo: eleven sequences as one job over two ranks"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, ("--sequences", "kitti-lengths", "--scale", "0.003")))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert outs[1].strip() == ""
    line = json.loads(outs[0].strip().splitlines()[-1])
    frames = [max(2, int(round(n * 0.003))) for n in (4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201)]
    assert line["n_gpus"] == 2 and line["steps"] == sum(frames) - 11 and line["config"]["trajectories"] == 11
    assert sum(line["config"]["pairs_per_rank"]) == line["steps"] and abs(line["config"]["pairs_per_rank"][0] - line["config"]["pairs_per_rank"][1]) <= 1
    assert line["config"]["ranks_seen"] == 2 and line["value"] > 0 and "config 3" in line["config"]["workload"]


def test_bench_gpus_flag_self_launch(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main(["--gpus", "8", "--steps", "3"])
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_committed_profile_feeds_the_hbm_roofline_entry():
    """roofline.hbm.other_kernels comes from the committed rocprofv3 passes (hardware counters cannot be read from inside the
    process): the files bench.py names must exist, parse, and price the streaming kernels below the 8 TB/s peak"""
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.exists(os.path.join(root, "profiles", bench.PMC_FILE)) and os.path.exists(os.path.join(root, "profiles", bench.STATS_FILE))
    ks = bench.committed_streaming_kernels()
    assert ks and all(0.0 < k["frac"] < 1.0 and k["mb_per_dispatch"] >= 4.0 for k in ks)
    names = {k["kernel"] for k in ks}
    assert {"k_warp", "k_correlation_rt", "k_deconv_dw4_blk"} <= names, names


def test_other_legs_need_a_gpu_and_never_break_the_line(monkeypatch):
    import bench
    assert bench.gpu_visible() is False  # (this container; on the GPU box /dev/kfd exists)
    r = bench.other_leg(["--definitely-not-an-option"], timeout=120)
    assert "error" in r and "argv" in r  # a failed leg is reported inside the line, the headline still prints


def test_mirror_options_precedence(monkeypatch):
    """DeepModel.initialize_models: f16x3 and the frame session unless the optional cfg key `dfvo_hip` or the environment says
    otherwise (environment first)"""
    import importlib
    import __graft_entry__ as g
    g.dfvo_amd()
    dm = importlib.import_module("df-vo_amd.libs.deep_models.deep_models")
    monkeypatch.delenv("DFVO_CONV_PRECISION", raising=False)
    monkeypatch.delenv("DFVO_SESSION", raising=False)
    assert dm.hip_options({}) == {"conv_precision": "f16x3", "session": True}
    assert dm.hip_options({"dfvo_hip": {"conv_precision": "fp32", "session": False}}) == {"conv_precision": "fp32", "session": False}
    monkeypatch.setenv("DFVO_CONV_PRECISION", "f16")
    monkeypatch.setenv("DFVO_SESSION", "0")
    assert dm.hip_options({"dfvo_hip": {"conv_precision": "fp32"}}) == {"conv_precision": "f16", "session": False}

    class Attr:  # an attribute-style config without the key (EasyDict raises AttributeError)
        pass
    monkeypatch.delenv("DFVO_CONV_PRECISION")
    monkeypatch.delenv("DFVO_SESSION")
    assert dm.hip_options(Attr()) == {"conv_precision": "f16x3", "session": True}
