"""Frame-batch data parallelism: every rank tracks a contiguous chunk of frame
pairs on its own GPU with no activation exchange; ONE RCCL all-gather of the per-frame relative poses
(4x4 f64 = 128 B/frame) + a status word per frame, then the sequential prefix composition that
reproduces DFVO.update_global_pose (libs/dfvo.py:109-119) including the constant-motion fallback
(dfvo.py:157-161).  The collective is latency bound (KB per sequence), so it is issued once per chunk."""
import numpy as np


def chunk_bounds(n_pairs, world, rank):
    """contiguous chunk [lo, hi) of frame pairs for `rank` (first ranks take the remainder)"""
    q, r = divmod(n_pairs, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allgather_poses(rel, status, world, rank, dist=None, backend_device="cuda"):
    """rel [n_local,4,4] f64, status [n_local] int64 -> gathered [sum n,4,4+1] (pose | status) in rank order.
    Uneven chunks are padded to the longest one for the fixed-size all-gather and trimmed afterwards."""
    n_local = rel.shape[0]
    if dist is None or world == 1:
        out = np.zeros((n_local, 17))
        out[:, :16] = rel.reshape(n_local, 16)
        out[:, 16] = status
        return out
    import torch
    dev = backend_device if dist.get_backend() == "nccl" else "cpu"
    cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    counts = [int(c.item()) for c in cnts]
    nmax = max(counts)
    buf = torch.zeros((nmax, 17), dtype=torch.float64, device=dev)
    if n_local:
        buf[:n_local, :16] = torch.from_numpy(rel.reshape(n_local, 16)).to(dev)
        buf[:n_local, 16] = torch.from_numpy(status.astype(np.float64)).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return np.concatenate([b[:c].cpu().numpy() for b, c in zip(bufs, counts)], 0)


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
