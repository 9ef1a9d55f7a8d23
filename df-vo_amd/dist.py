"""Frame-batch data parallelism: every rank tracks a contiguous chunk of frame
pairs on its own GPU with no activation exchange; ONE RCCL all-gather of the per-frame relative poses
(4x4 f64 = 128 B/frame) + a status word per frame, then the sequential prefix composition that
reproduces DFVO.update_global_pose (libs/dfvo.py:109-119) including the constant-motion fallback
(dfvo.py:157-161).  The collective is latency bound (KB per sequence), so it is issued ONCE PER JOB: the chunking
(chunk_bounds for one sequence, job_items for several) is deterministic, every rank knows every rank's row count."""
import numpy as np


def chunk_bounds(n_pairs, world, rank):
    """contiguous chunk [lo, hi) of frame pairs for `rank` (first ranks take the remainder)"""
    q, r = divmod(n_pairs, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def job_items(seq_pairs, world):
    """Work list of a multi-sequence job (BASELINE config 3: KITTI 00-10 frame-batched across the GPUs of a node).
    The frame pairs of all sequences are laid end to end and cut into `world` contiguous ranges of equal length (+-1, as
    chunk_bounds does), each range split where it crosses a sequence boundary: per rank a list of items (sequence index,
    lo, hi) over that sequence's pair indices.  Every item starts from its own 1-frame halo (sequence.track_chunk), so no
    activation crosses ranks or items; the split is a pure function of (seq_pairs, world), so every rank knows every
    rank's row count and no count exchange precedes the all-gather."""
    starts = np.concatenate([[0], np.cumsum(np.asarray(seq_pairs, np.int64))])
    total = int(starts[-1])
    items = []
    for rank in range(world):
        lo, hi = chunk_bounds(total, world, rank)
        mine = []
        for s in range(len(seq_pairs)):
            a, b = max(lo, int(starts[s])), min(hi, int(starts[s + 1]))
            if b > a:
                mine.append((s, a - int(starts[s]), b - int(starts[s])))
        items.append(mine)
    return items


class RcclComm:
    """the C ABI's communicator (include/dfvo_hip.h dfvo_comm_*: one RCCL communicator, ncclAllGather on its own HIP stream).
    `bcast_id(bytes or None) -> bytes` ships rank 0's unique id to every rank out of band; from_torch() uses the process
    group the host already owns for that (one object broadcast at start-up; the pose exchange itself then runs through
    the C ABI, the path a non-Python host takes)."""

    def __init__(self, world, rank, bcast_id):
        import ctypes as C
        from . import capi
        self._capi, self._C = capi, C
        lib = capi.lib()
        idb = (C.c_uint8 * 128)()
        if rank == 0:
            capi.check(lib.dfvo_comm_unique_id(idb))
        raw = bcast_id(bytes(idb) if rank == 0 else None)
        idb = (C.c_uint8 * 128).from_buffer_copy(raw)
        self.h = C.c_void_p()
        capi.check(lib.dfvo_comm_create(idb, world, rank, C.byref(self.h)))
        self.world, self.rank = world, rank

    @classmethod
    def from_torch(cls, dist, world, rank):
        def bcast(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return cls(world, rank, bcast)

    def allgather_rows(self, rows, counts):
        C, capi = self._C, self._capi
        rows = np.ascontiguousarray(rows, np.float64).reshape(-1, 17)
        out = np.zeros((int(sum(counts)), 17))
        cnt = (C.c_int * self.world)(*[int(c) for c in counts])
        capi.check(capi.lib().dfvo_allgather_poses(self.h, capi.as_ptr(rows) if len(rows) else None, len(rows), cnt, capi.as_ptr(out)))
        return out

    def close(self):
        if self.h:
            self._capi.lib().dfvo_comm_destroy(self.h)
            self.h = None


def pack_rows(rel, status):
    """rel [n,4,4] f64, status [n] -> rows [n,17] (pose | status): the layout of the collective and of dfvo_compose_trajectory"""
    n = rel.shape[0]
    out = np.zeros((n, 17))
    out[:, :16] = rel.reshape(n, 16)
    out[:, 16] = status
    return out


def allgather_rows(rows, counts, world, rank, dist=None, comm=None, backend_device="cuda"):
    """ONE collective: rows [counts[rank],17] of every rank -> [sum(counts),17] in rank order.  `counts` is known to every
    rank beforehand (chunk_bounds / job_items are deterministic), ranks pad to max(counts) for the fixed-size all-gather.
    comm (RcclComm): the C ABI's dfvo_allgather_poses; else torch.distributed (nccl = RCCL on device tensors, gloo on host
    tensors in the CPU tests)."""
    rows = np.ascontiguousarray(rows, np.float64).reshape(-1, 17)
    if len(counts) != world or rows.shape[0] != counts[rank]:
        raise ValueError("allgather_rows: counts must list every rank's row count (this rank: %d rows)" % rows.shape[0])
    if world == 1:
        return rows.copy()
    if comm is not None:
        return comm.allgather_rows(rows, counts)
    if dist is None:
        raise ValueError("allgather_rows: world > 1 needs a process group or an RcclComm")
    import torch
    nmax = int(max(counts))
    nccl = dist.get_backend() == "nccl"
    dev = backend_device if nccl else "cpu"
    buf = torch.zeros((nmax, 17), dtype=torch.float64, device=dev)
    if rows.shape[0]:
        buf[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    if nccl:
        out = torch.empty((world * nmax, 17), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, buf)
        bufs = out.view(world, nmax, 17).cpu()
    else:
        bufs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
    return np.concatenate([bufs[r][:counts[r]].numpy() for r in range(world)], 0)


def allgather_poses(rel, status, world, rank, dist=None, counts=None, comm=None, backend_device="cuda"):
    """rel [n_local,4,4] f64, status [n_local] int64 -> gathered [sum n,17] (pose | status) in rank order: pack_rows +
    allgather_rows.  counts: every rank's n_local (required when world > 1; sequence.run_sequence derives it from
    chunk_bounds -- there is no count exchange)."""
    rows = pack_rows(rel, status)
    if world == 1 or (dist is None and comm is None):
        return rows
    if counts is None:
        raise ValueError("allgather_poses: pass counts (the chunking is deterministic; no count exchange is made)")
    return allgather_rows(rows, list(counts), world, rank, dist, comm, backend_device)


def compose_trajectory_device(gathered, first_pose=None):
    """compose_trajectory in ONE device launch (dfvo_compose_trajectory[_device], csrc/solver_pipeline.hip): `gathered` is
    the [n,17] pose | status array -- a numpy array, or a CUDA tensor (the RCCL all-gather's output: nothing but the
    composed poses then crosses PCIe).  Same recurrence, same order of operations as the host loop below; numpy's 3x3
    products may fuse multiply-adds where the kernel does not, so the two agree to rounding (<= 1e-12), not bit for bit."""
    import ctypes as C
    from . import capi
    lib = capi.lib()
    bad = C.c_int(-1)
    fp = None if first_pose is None else np.ascontiguousarray(first_pose, np.float64)
    if hasattr(gathered, "is_cuda") and gathered.is_cuda:
        import torch
        g = gathered.contiguous().double()
        n = g.shape[0]
        out = torch.empty((n + 1, 4, 4), dtype=torch.float64, device=g.device)
        first = None if fp is None else torch.from_numpy(fp).to(g.device)
        torch.cuda.current_stream().synchronize()  # the library launches on the null stream
        capi.check(lib.dfvo_compose_trajectory_device(C.c_void_p(g.data_ptr()), n, None if first is None else C.c_void_p(first.data_ptr()),
                                                      C.c_void_p(out.data_ptr()), C.byref(bad), None))
        poses = out.cpu().numpy()
    else:
        g = np.ascontiguousarray(gathered, np.float64)
        n = g.shape[0]
        poses = np.zeros((n + 1, 4, 4))
        capi.check(lib.dfvo_compose_trajectory(capi.as_ptr(g), n, None if fp is None else capi.as_ptr(fp), capi.as_ptr(poses), C.byref(bad)))
    if bad.value >= 0:
        raise ValueError("compose_trajectory: pair %d needs the PnP fallback but had no reference depth; start every chunk "
                         "with TrackingPipeline.set_ref_image(first frame of the chunk)" % bad.value)
    return poses


def compose_trajectory(gathered, first_pose=None):
    """sequential prefix composition over all gathered pairs -> global poses [n+1,4,4].
    status 1 (constant motion) reuses the previous pair's relative motion, as the reference does."""
    n = gathered.shape[0]
    bad = np.nonzero(gathered[:, 16] == 2)[0]
    if bad.size:  # DFVO_TRACK_NEEDS_PNP: the row holds no pose (the chunk started without a reference depth)
        raise ValueError("compose_trajectory: pair(s) %s need the PnP fallback but had no reference depth; start every "
                         "chunk with TrackingPipeline.set_ref_image(first frame of the chunk)" % bad[:8].tolist())
    g = np.eye(4) if first_pose is None else first_pose.copy()
    out = np.zeros((n + 1, 4, 4))
    out[0] = g
    prev = np.eye(4)
    for i in range(n):
        rel = gathered[i, :16].reshape(4, 4)
        if int(gathered[i, 16]) == 1:
            rel = prev
        nxt = g.copy()
        nxt[:3, 3:] = g[:3, :3] @ rel[:3, 3:] + g[:3, 3:]
        nxt[:3, :3] = g[:3, :3] @ rel[:3, :3]
        g = nxt
        prev = rel
        out[i + 1] = g
    return out
