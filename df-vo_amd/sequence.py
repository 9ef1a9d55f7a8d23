"""Sequence driver over the fused pipeline: the analogue of DFVO.main's frame loop (/root/reference/libs/dfvo.py:347-425)
for one contiguous chunk of a sequence, and the frame-batch data-parallel wrapper around it.

Pair j = (frame j, frame j+1).  A chunk [lo, hi) of pairs needs frames lo .. hi: frame lo is the chunk's 1-frame halo
(its depth is the PnP fallback's reference depth for pair lo; the reference computes it when frame lo is `cur`, here it
is recomputed by the rank that owns the chunk -- no activation exchange between ranks).  The nets run `ahead` pairs
ahead of the solver stage; nothing but the relative poses leaves the device.

RandomState modes (SURVEY hard part 2):
  "sequential"  one stream seeded once (np.random.seed(cfg.seed), run.py:81-84) and carried from pair to pair -- the
                reference's behaviour, reproducible only by a single rank walking the whole sequence in order;
  "per_pair"    the stream is re-seeded with seed ^ (pair index + 1) before every pair, so a pair's result does not depend
                on which rank tracked it or what came before: the data-parallel mode (documented deviation)."""
import numpy as np
import torch

from . import capi
from . import dist as dmod
from .pipeline import TrackingPipeline

SLOTS = 4  # DFVO_PIPELINE_SLOTS


def track_chunk(pipe, frames, lo, hi, seed=4869, rng_mode="sequential", ahead=3, first_chunk=None, collect=None,
                carry_features=True):
    """track pairs lo .. hi-1 of `frames` (sequence of device uint8 tensors [H,W,3], indexable by frame number).
    Returns (rel [n,4,4] cur->ref motions, status [n]); status-1 rows (constant motion) hold the identity and are resolved
    by dist.compose_trajectory once the previous pair's motion is known."""
    assert rng_mode in ("sequential", "per_pair")
    n = hi - lo
    rel = np.tile(np.eye(4), (n, 1, 1))
    status = np.zeros(n, np.int64)
    if n <= 0:
        return rel, status
    ahead = max(1, min(ahead, SLOTS - 1))
    pipe.set_ref_image(frames[lo])  # halo: depth of the chunk's first reference frame
    if rng_mode == "sequential":
        pipe.seed(seed)
    # carry_features: only the chunk's first pair hands both frames to the flow net; from then on the reference frame's image
    # / feature pyramids are the ones the previous pair computed for it as its current frame (ref = None)

    def feed(j):
        pipe.enqueue_nets(j % SLOTS, None if (carry_features and j > lo) else frames[j], frames[j + 1])
        pipe.prefetch_track(j % SLOTS)

    for j in range(lo, min(lo + ahead, hi)):
        feed(j)
    prev = np.eye(4)
    for j in range(lo, hi):
        if rng_mode == "per_pair":
            pipe.seed((seed ^ (j + 1)) & 0xffffffff)
        pipe.track_begin(j % SLOTS)  # the chain of pair j runs while the host feeds the nets of pair j + ahead
        if j + ahead < hi:
            feed(j + ahead)
        out = pipe.track_end(j % SLOTS)
        status[j - lo] = out.status
        if out.status != 1:
            rel[j - lo], _ = TrackingPipeline.hybrid_pose(out, prev)
            prev = rel[j - lo]
        if collect is not None:
            collect(j, out)
    pipe.sync()
    return rel, status


def run_sequence(pipe, frames, n_frames, world=1, rank=0, dist=None, seed=4869, rng_mode=None, ahead=3, collect=None,
                 compose="host", carry_features=True, comm=None):
    """data-parallel tracking of an n_frames sequence: contiguous chunk per rank (dist.chunk_bounds), ONE all-gather of
    the relative poses + status words (RCCL when `dist` runs the nccl backend), then the sequential prefix composition
    that reproduces DFVO.update_global_pose incl. the constant-motion rule (dfvo.py:109-119,157-161).
    compose: "host" = the numpy recurrence of dist.compose_trajectory (the reference's own operations), "device" = the same
    recurrence in one launch (dist.compose_trajectory_device; agrees to rounding).
    Returns (global poses [n_frames,4,4] camera-to-world, gathered [n_frames-1,17])."""
    if rng_mode is None:
        rng_mode = "sequential" if world == 1 else "per_pair"
    if world > 1 and rng_mode == "sequential":
        raise ValueError("the sequential numpy RandomState cannot be reproduced frame-parallel; use rng_mode='per_pair'")
    bounds = [dmod.chunk_bounds(n_frames - 1, world, r) for r in range(world)]
    lo, hi = bounds[rank]
    rel, status = track_chunk(pipe, frames, lo, hi, seed, rng_mode, ahead, collect=collect, carry_features=carry_features)
    gathered = dmod.allgather_poses(rel, status, world, rank, dist, counts=[b - a for a, b in bounds], comm=comm)
    traj = dmod.compose_trajectory_device(gathered) if compose == "device" else dmod.compose_trajectory(gathered)
    return traj, gathered


def run_sequences(pipe, seqs, world=1, rank=0, dist=None, comm=None, seed=4869, rng_mode=None, ahead=3, compose="host",
                  carry_features=True, out_dir=None, gts=None, alignment=None, collect=None):
    """BASELINE config 3 (KITTI 00-10 frame-batched across the GPUs of one node): what the reference does as eleven
    runs of apis/run.py, one `--seq` each (DFVO.main, /root/reference/libs/dfvo.py:347-425; trajectory written by
    save_traj, libs/general/utils.py:329-355; evaluated by tools/evaluation/odometry/eval_odom.py).

    seqs: list of (name, frames, n_frames) -- `frames` indexable by frame number -> device uint8 [H,W,3].
    The pairs of all sequences form one work list balanced over the ranks by frame count (dist.job_items); every item
    starts from its own 1-frame halo; ONE all-gather for the whole job (row counts are deterministic, nothing else is
    exchanged); then every sequence is composed separately from the identity (a fresh DFVO per sequence), written to
    out_dir/<name>.txt by rank 0 and, when gts[name] ([n,4,4]) is given, evaluated with evaluation.evaluate(alignment).
    RandomState: world 1 -> "sequential", re-seeded at the start of every sequence as run.py does per run; world > 1 ->
    "per_pair" keyed by the pair's index within its sequence (a pair's result does not depend on the job's composition).
    Returns {name: {"poses": [n,4,4], "gathered": [n-1,17], "metrics": dict or None}} in the order of `seqs`."""
    from . import evaluation as ev
    if rng_mode is None:
        rng_mode = "sequential" if world == 1 else "per_pair"
    if world > 1 and rng_mode == "sequential":
        raise ValueError("the sequential numpy RandomState cannot be reproduced frame-parallel; use rng_mode='per_pair'")
    n_pairs = [max(0, int(n) - 1) for _, _, n in seqs]
    items = dmod.job_items(n_pairs, world)
    rows = []
    for s, lo, hi in items[rank]:
        name, frames, _ = seqs[s]
        rel, status = track_chunk(pipe, frames, lo, hi, seed, rng_mode, ahead, carry_features=carry_features,
                                  collect=None if collect is None else (lambda j, out, _n=name: collect(_n, j, out)))
        rows.append(dmod.pack_rows(rel, status))
    mine = np.concatenate(rows, 0) if rows else np.zeros((0, 17))
    counts = [sum(hi - lo for _, lo, hi in it) for it in items]
    gathered = dmod.allgather_rows(mine, counts, world, rank, dist, comm)
    out = {}
    off = 0
    for (name, _, n), npair in zip(seqs, n_pairs):
        g = gathered[off:off + npair]
        off += npair
        poses = dmod.compose_trajectory_device(g) if compose == "device" else dmod.compose_trajectory(g)
        metrics = None
        if gts is not None and gts.get(name) is not None:
            metrics = ev.evaluate(np.asarray(gts[name])[:len(poses)], poses, alignment=alignment)
        if out_dir is not None and rank == 0:
            import os
            os.makedirs(out_dir, exist_ok=True)
            ev.save_traj(os.path.join(out_dir, "%s.txt" % name), poses)
        out[name] = {"poses": poses, "gathered": g, "metrics": metrics}
    return out


def save_traj(path, poses):
    """one line per frame: "<idx> r11 r12 r13 tx r21 ... tz" (libs/general/utils.py:329-355 save_traj, format 'kitti');
    the batched writer of df-vo_amd/evaluation.py"""
    from .evaluation import save_traj as _save
    _save(path, poses)


def frames_to_device(frames_u8):
    return [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in frames_u8]


def image_to_device(img_u8, h, w, crop=None, bgr=False):
    """read_image (/root/reference/libs/general/utils.py:44-51) behind the decoder, on the device: the decoded frame is uploaded
    as it is -- bgr=True: as cv2.imread returns it, the BGR -> RGB conversion of utils.py:45 is folded into the read --,
    cropped ([[y0, y1], [x0, x1]] fractions, utils.py:46-50) and resized to the configured (h, w) with cv2.resize's default
    8-bit INTER_LINEAR arithmetic, in ONE launch (dfvo_read_image_tail_u8).  Returns a device uint8 tensor [h, w, 3] (RGB)."""
    img = np.ascontiguousarray(img_u8)
    assert img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8
    ih, iw = img.shape[:2]
    y0, y1, x0, x1 = 0, ih, 0, iw
    if crop is not None:
        y0, y1 = int(ih * crop[0][0]), int(ih * crop[0][1])
        x0, x1 = int(iw * crop[1][0]), int(iw * crop[1][1])
    src = torch.from_numpy(img).cuda()
    dst = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    capi.check(capi.lib().dfvo_read_image_tail_u8(src.data_ptr(), ih, iw, 1 if bgr else 0, y0, y1, x0, x1, dst.data_ptr(), h, w,
                                                  stream.cuda_stream))
    # Ordering contract for device frames handed to TrackingPipeline.enqueue_nets / set_ref_image: the pipeline reads them
    # on its OWN non-blocking streams, which are not ordered behind torch's current stream -- a frame must be complete
    # before it is handed over.  The upload + resize are therefore drained here (frames_to_device's .cuda() copy is
    # synchronous for pageable host memory already).
    stream.synchronize()
    return dst
