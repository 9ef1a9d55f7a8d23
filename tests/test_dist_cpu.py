"""CPU (gloo, world_size 2 and 3): the frame-batch data-parallel driver -- chunking, the single pose
all-gather (uneven chunks), and the prefix composition with the constant-motion rule."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _poses(n, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed))
    rel = np.tile(np.eye(4), (n, 1, 1))
    for i in range(n):
        a = rng.normal(0, 0.01, 3)
        th = np.linalg.norm(a)
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        rel[i, :3, :3] = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        rel[i, :3, 3] = rng.normal(0, 0.5, 3)
    status = (rng.random(n) < 0.15).astype(np.int64)  # some constant-motion frames
    status[0] = 0
    return rel, status


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    dmod = importlib.import_module("df-vo_amd.dist")
    rel, status = _poses(n)
    lo, hi = dmod.chunk_bounds(n, world, rank)
    counts = [b - a for a, b in (dmod.chunk_bounds(n, world, r) for r in range(world))]  # deterministic: no count exchange
    g = dmod.allgather_poses(rel[lo:hi], status[lo:hi], world, rank, dist, counts=counts)
    if rank == 0:
        q.put(g)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    g = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return g


def test_allgather_and_compose_world2_and_3():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    dmod = importlib.import_module("df-vo_amd.dist")
    for world, n in ((2, 11), (3, 10)):
        rel, status = _poses(n)
        g = _run(world, n)
        assert g.shape == (n, 17)
        assert np.array_equal(g[:, :16].reshape(n, 4, 4), rel) and np.array_equal(g[:, 16].astype(np.int64), status)
        traj = dmod.compose_trajectory(g)
        # sequential reference: dfvo.py:109-119 with the constant-motion rule of dfvo.py:157-161
        cur = np.eye(4)
        prev = np.eye(4)
        for i in range(n):
            r = prev if status[i] == 1 else rel[i]
            t = cur[:3, :3] @ r[:3, 3:] + cur[:3, 3:]
            R = cur[:3, :3] @ r[:3, :3]
            cur = np.eye(4)
            cur[:3, :3], cur[:3, 3:] = R, t
            prev = r
            assert np.allclose(traj[i + 1], cur, atol=0, rtol=0)


def test_chunk_bounds_cover_everything():
    dmod = importlib.import_module("df-vo_amd.dist")
    for n in (0, 1, 7, 1590):
        for world in (1, 2, 3, 8):
            spans = [dmod.chunk_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]


# ---------------------------------------------------------------------------------------------------------------
# sequence.run_sequence (chunk + 1-frame halo driver) under gloo, the device pipeline replaced by a host stub whose
# result for a pair depends only on (the two frames, the RandomState seed in force) -- as the real pipeline's does
# ---------------------------------------------------------------------------------------------------------------
class _StubOut:
    pass


class _StubPipe:
    def __init__(self):
        self.slots, self.seed_, self.ref_image, self.log = {}, None, None, []
        self.last_cur = None

    def set_ref_image(self, f):
        self.ref_image = int(f)
        self.log.append(("ref", int(f)))

    def seed(self, s):
        self.seed_ = int(s)

    def enqueue_nets(self, slot, ref, cur, feed=None):
        if ref is None:  # carried mode: the reference frame is the previous call's current frame
            assert self.last_cur is not None
            ref = self.last_cur
        assert int(cur) == int(ref) + 1
        self.last_cur = int(cur)
        self.slots[slot] = (int(ref), int(cur))

    def prefetch_track(self, slot, *a):
        pass

    def track(self, slot, *a):
        ref, cur = self.slots[slot]
        assert self.ref_image is not None and self.ref_image <= ref  # the chunk's halo frame came first
        rng = np.random.Generator(np.random.PCG64([ref, cur, self.seed_]))
        o = _StubOut()
        o.status = 1 if rng.random() < 0.25 else (3 if rng.random() < 0.2 else 0)
        a = rng.normal(0, 0.02, 3)
        th = np.linalg.norm(a)
        k = a / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        o.R = list((np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).reshape(-1))
        o.t = list(rng.normal(0, 1, 3))
        o.scale = float(rng.uniform(0.5, 1.5))
        return o

    def track_begin(self, slot, *a):  # (the result of a pair is fixed by what is in force when its chain is enqueued)
        self.pending = getattr(self, "pending", {})
        self.pending[slot] = self.track(slot)

    def track_end(self, slot):
        return self.pending.pop(slot)

    def sync(self):
        pass


def _seq_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    smod = importlib.import_module("df-vo_amd.sequence")
    pipe = _StubPipe()
    poses, gathered = smod.run_sequence(pipe, list(range(n_frames)), n_frames, world, rank, dist, seed=4869)
    if rank == 0:
        q.put((poses, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_run_sequence_chunks_equal_single_rank_per_pair_seed():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    smod = importlib.import_module("df-vo_amd.sequence")
    n_frames = 24
    want_poses, want_g = smod.run_sequence(_StubPipe(), list(range(n_frames)), n_frames, 1, 0, None, seed=4869,
                                           rng_mode="per_pair")
    assert (want_g[:, 16] == 1).any() and (want_g[:, 16] == 3).any()  # constant-motion and PnP rows are exercised
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_seq_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
        for p in procs:
            p.start()
        poses, g = q.get(timeout=120)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert np.array_equal(g, want_g) and np.array_equal(poses, want_poses)
    # the sequential numpy stream cannot be chunked; and a needs-PnP row must not be composed silently
    import pytest
    with pytest.raises(ValueError):
        smod.run_sequence(_StubPipe(), list(range(6)), 6, 2, 0, None, rng_mode="sequential")
    dmod = importlib.import_module("df-vo_amd.dist")
    bad = want_g.copy()
    bad[3, 16] = 2
    with pytest.raises(ValueError):
        dmod.compose_trajectory(bad)


# ---------------------------------------------------------------------------------------------------------------
# sequence.run_sequences: BASELINE config 3 (eleven sequences frame-batched over the ranks, one all-gather for the job)
# ---------------------------------------------------------------------------------------------------------------
KITTI_FRAMES = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201]  # gt_poses/00..10.txt line counts


def _job(scale=0.004):
    """eleven stub sequences with KITTI's length ratios; frame ids are unique per sequence (1000 * k + i)"""
    seqs = []
    for k, n in enumerate(KITTI_FRAMES):
        nf = max(2, int(round(n * scale)))
        seqs.append(("%02d" % k, [1000 * k + i for i in range(nf)], nf))
    return seqs


class _CountingDist:
    """wraps torch.distributed: counts the collectives run_sequences issues"""

    def __init__(self, d):
        self.d, self.calls = d, []

    def get_backend(self):
        return self.d.get_backend()

    def all_gather(self, *a, **k):
        self.calls.append("all_gather")
        return self.d.all_gather(*a, **k)

    def __getattr__(self, name):
        def f(*a, **k):
            self.calls.append(name)
            return getattr(self.d, name)(*a, **k)
        return f


def _job_worker(rank, world, port, out_dir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    smod = importlib.import_module("df-vo_amd.sequence")
    cd = _CountingDist(dist)
    pipe = _StubPipe()
    res = smod.run_sequences(pipe, _job(), world, rank, cd, seed=4869, out_dir=out_dir)
    if rank == 0:
        q.put(({k: (v["poses"], v["gathered"]) for k, v in res.items()}, cd.calls, [e for e in pipe.log if e[0] == "ref"]))
    dist.barrier()
    dist.destroy_process_group()


def test_job_items_balance_and_cover_kitti_lengths():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dmod = importlib.import_module("df-vo_amd.dist")
    pairs = [n - 1 for n in KITTI_FRAMES]
    assert sum(KITTI_FRAMES) == 23201
    for world in (1, 2, 3, 4, 8):
        items = dmod.job_items(pairs, world)
        loads = [sum(hi - lo for _, lo, hi in it) for it in items]
        assert sum(loads) == sum(pairs) and max(loads) - min(loads) <= 1
        cover = {s: [] for s in range(len(pairs))}
        for it in items:
            for s, lo, hi in it:
                assert 0 <= lo < hi <= pairs[s]
                cover[s].append((lo, hi))
        for s, spans in cover.items():  # every sequence covered exactly once, in order
            spans.sort()
            assert spans[0][0] == 0 and spans[-1][1] == pairs[s]
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert dmod.job_items([0, 3], 2) == [[(1, 0, 2)], [(1, 2, 3)]]  # a one-frame sequence has no pair


def test_run_sequences_world2_and_3_equal_single_rank(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    smod = importlib.import_module("df-vo_amd.sequence")
    ev = importlib.import_module("df-vo_amd.evaluation")
    seqs = _job()
    want = smod.run_sequences(_StubPipe(), seqs, 1, 0, None, seed=4869, rng_mode="per_pair")
    assert list(want) == ["%02d" % k for k in range(11)]
    assert any((v["gathered"][:, 16] == 1).any() for v in want.values())
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        out_dir = str(tmp_path / ("w%d" % world))
        procs = [ctx.Process(target=_job_worker, args=(r, world, port, out_dir, q)) for r in range(world)]
        for p in procs:
            p.start()
        got, calls, refs = q.get(timeout=120)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert calls.count("all_gather") == 1 and [c for c in calls if c not in ("all_gather", "get_backend")] == []  # ONE collective
        for name, v in want.items():
            assert np.array_equal(got[name][0], v["poses"]) and np.array_equal(got[name][1], v["gathered"])
            assert np.array_equal(ev.load_traj(os.path.join(out_dir, name + ".txt")), v["poses"])  # eleven trajectory files
        assert len(refs) >= 1  # every item of rank 0 started from its own halo frame
    # sequential mode (one rank): every sequence re-seeds, i.e. equals that sequence tracked alone
    seq_all = smod.run_sequences(_StubPipe(), seqs, 1, 0, None, seed=4869)
    for name, frames, n in seqs[:3]:
        alone, _ = smod.run_sequence(_StubPipe(), frames, n, 1, 0, None, seed=4869)
        assert np.array_equal(seq_all[name]["poses"], alone)
    # metrics per sequence when ground truth is given
    gts = {name: want[name]["poses"] for name, _, _ in seqs}
    ev_out = smod.run_sequences(_StubPipe(), seqs, 1, 0, None, seed=4869, rng_mode="per_pair", gts=gts, alignment="6dof")
    assert all(abs(v["metrics"]["ate"]) < 1e-9 for v in ev_out.values())
