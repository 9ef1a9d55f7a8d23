"""Frame session under the drop-in classes: the reference's own synchronous call order (/root/reference/libs/dfvo.py:299-345,
121-262), fast, with every call returning exactly what it returned before (C side: df-vo_amd/csrc/session.hip).

DeepModel.forward_depth(k) PUSHES frame k: one upload, the depth net of k and the flow net of (k - 1, k) on their own
streams, and -- once KeypointSampler / EssTracker have registered their configurations (libs/tracker/_ctx.py) -- the keypoint
selection and the RandomState-independent half of compute_pose_2d2d behind the flow net.  The later calls of the frame loop
wait for an event and hand out the result, IF they are asked for what was enqueued:
  * forward_flow compares the two frames it is given, byte for byte, with the session's pinned copies of what was uploaded,
  * the flow / consistency arrays it returns are SessionArray views of pinned host buffers; a copy of one (dfvo.py:330-333
    copies them) keeps the generation token, a write through the array object drops it, and kp_selection compares the WHOLE
    contents with the session's buffer before it trusts the device-resident copy (a write through a child view, np.copyto,
    cv2 with dst= ... is caught there),
  * compute_pose_2d2d compares keypoints and configuration byte for byte on the C side; its RandomState-consuming half is
    enqueued AHEAD, from forward_flow, under np.random's state at that moment, and its result is handed out only if
    np.random's state at the compute_pose_2d2d call is still that state, word for word (DFVO_SESSION_POSE_AHEAD=0: never ahead).
Anything else takes the plain host-array entry point -- same results, the round-3 speed.

Lifetime of what the calls return.  forward_depth / forward_flow return READ-ONLY views of the session's pinned ring (three
slots).  Every view keeps its slot's lease alive (ndarray.base chain); when a slot comes up for reuse -- the push of the third
frame after it -- while any view, or view of a view, of it is still referenced, the session hands the slot's buffers over to
the lease (dfvo_session_detach_slot: the session allocates itself fresh ones, the old ones are freed when the last view dies)
-- likewise at close().  A returned array therefore never changes under its holder and never dangles; what differs from the
reference's owned arrays is that it is read-only (write to a .copy(), as dfvo.py:330-333 does).

Arithmetic guard.  With an f16x3 / f16 packing an activation beyond +-65504 becomes inf in the layer that splits it; the
device counter of such events is read behind each net (4 bytes riding the output copy) and forward_depth / forward_flow
raise capi.DfvoError instead of returning inf / NaN maps."""
import ctypes as C
import os
import weakref

import numpy as np

from ... import capi

RING = 3  # host buffer sets of the C side: the push of frame g + 3 writes the slot of frame g (see "Lifetime" above)


class SessionArray(np.ndarray):
    """ndarray view / copy that remembers which session buffer (session id, generation, name) its contents came from"""
    _dfvo_tok = None

    def __array_finalize__(self, obj):
        tok = getattr(obj, "_dfvo_tok", None)
        same = obj is not None and tok is not None and self.shape == obj.shape and self.dtype == obj.dtype
        self._dfvo_tok = tok if same else None

    def __setitem__(self, key, value):
        self._dfvo_tok = None
        super().__setitem__(key, value)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        plain = tuple(np.asarray(x) if isinstance(x, SessionArray) else x for x in inputs)
        if out is None:
            return getattr(ufunc, method)(*plain, **kwargs)  # results are plain arrays
        for o in out:
            if isinstance(o, SessionArray):
                o._dfvo_tok = None  # written in place: no longer the session's contents
        pout = tuple(np.asarray(o) if isinstance(o, SessionArray) else o for o in out)
        res = getattr(ufunc, method)(*plain, out=pout, **kwargs)
        back = {id(p): o for p, o in zip(pout, out)}  # in-place operators must hand the object itself back
        if isinstance(res, tuple):
            return tuple(back.get(id(r), r) for r in res)
        return back.get(id(res), res)


_libc = C.CDLL(None)
_libc.memcmp.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
_libc.memcmp.restype = C.c_int


def same_bytes(a, b):
    """whole-array comparison, byte for byte (NaNs compare by their bits), of two arrays of one shape and dtype"""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.flags["C_CONTIGUOUS"] and b.flags["C_CONTIGUOUS"]:
        return _libc.memcmp(a.ctypes.data, b.ctypes.data, a.nbytes) == 0
    return np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes()


def _free_pinned(ptrs):
    lib = capi.lib()
    for p in ptrs:
        lib.dfvo_host_free(p)


class _Lease:
    """one generation's hold on its ring slot: alive as long as any array handed out for that generation is"""

    def __init__(self, gen):
        self.gen = gen

    def adopt(self, ptrs):
        """the slot was reused / the session closed while views were alive: the buffers are this lease's now"""
        weakref.finalize(self, _free_pinned, [int(p) for p in ptrs if p])


class _Block:
    """a typed window onto pinned memory that numpy can wrap (array interface); keeps the lease alive through ndarray.base"""

    def __init__(self, ptr, typestr, shape, lease):
        self.lease = lease
        self.__array_interface__ = {"version": 3, "data": (int(ptr), True), "typestr": typestr, "shape": tuple(shape)}


def _pinned(ptr, ctype, shape, lease=None):
    typestr = {C.c_float: "<f4", C.c_double: "<f8", C.c_ubyte: "|u1"}[ctype]
    return np.asarray(_Block(ptr.value if hasattr(ptr, "value") else ptr, typestr, shape, lease))


class FrameSession:
    _next_id = 0
    IDLE_PUSHES = 3  # pushes in a row whose flow nobody asked for, after which a push stops enqueuing the flow pass

    def __init__(self, flow_model, depth_model, tracker, height, width, feed_height, feed_width):
        self.h, self.w, self.fh, self.fw = int(height), int(width), int(feed_height), int(feed_width)
        self.lib = capi.lib()
        self._keep = (flow_model, depth_model, tracker)
        h = C.c_void_p()
        capi.check(self.lib.dfvo_session_create(flow_model, depth_model, tracker, self.h, self.w, C.byref(h)))
        self.handle = h
        FrameSession._next_id += 1
        self.sid = FrameSession._next_id
        self.active = True      # False: another session took over the tracker (libs/tracker/_ctx.register_session) -- plain paths only
        self.gen = -1
        self.leases = {}        # generation -> weakref to the _Lease of the arrays handed out for it
        self.flow_views = None  # (generation, fwd, bwd, diff [H,W,1]) of the newest pair
        self.kp_cfg = None      # capi.SessionKpCfg registered by KeypointSampler (None: no speculative selection)
        self.pose_cfg_fn = None  # () -> capi.Pose2d2dCfg registered by EssTracker
        self.kp_spec = None     # the kp cfg the newest push ran with
        self.spec_inflight = False  # a speculative keypoint / homography stage may be running on the tracker's buffers
        self.pose_ahead = os.environ.get("DFVO_SESSION_POSE_AHEAD", "1") != "0"
        self.pose_early = os.environ.get("DFVO_SESSION_POSE_EARLY", "1") != "0"  # ... before the flow copies are waited for
        self.rng_words = None   # () -> np.random's state as 625 uint32 words (libs/tracker/_ctx.numpy_rng_words)
        self.have_flow = False  # the newest push enqueued the flow pass of (gen - 1, gen)
        self.unconsumed = 0     # pushes since forward_flow last asked for a pair (a depth-only caller stops paying for the flow net)
        self.stats = {"push": 0, "push_no_flow": 0, "flow_resident": 0, "flow_plain": 0, "kp_resident": 0, "kp_plain": 0,
                      "pose_resident": 0, "pose_plain": 0, "pose_ahead": 0, "slots_detached": 0}

    def _detach(self, gen):
        """the arrays handed out for `gen` are still referenced while their ring slot is needed again (or the session goes
        away): they keep the memory, the session gets fresh buffers"""
        ref = self.leases.pop(gen, None)
        lease = ref() if ref is not None else None
        if lease is None or self.handle is None:
            return
        old = (C.c_void_p * 4)()
        capi.check(self.lib.dfvo_session_detach_slot(self.handle, gen, old))
        lease.adopt(list(old))
        self.stats["slots_detached"] += 1

    def _lease(self, gen):
        ref = self.leases.get(gen)
        lease = ref() if ref is not None else None
        if lease is None:
            lease = _Lease(gen)
            self.leases[gen] = weakref.ref(lease)
        return lease

    def close(self):
        if self.handle is not None:
            for gen in list(self.leases):
                self._detach(gen)  # (views that outlive the session own their memory from here on)
            self.lib.dfvo_session_destroy(self.handle)
            self.handle = None
        self.active = False

    def retire(self):
        """another session was registered over the same tracker: this one stops speculating (its DeepModel keeps working
        through the plain entry points)"""
        if self.handle is not None and self.active:
            capi.check(self.lib.dfvo_session_quiesce(self.handle))
        self.active = False
        self.spec_inflight = False

    def reset(self):
        capi.check(self.lib.dfvo_session_reset(self.handle))
        for gen in list(self.leases):
            self._detach(gen)  # (generations restart at 0: the old numbering must not alias the new one)
        self.gen = -1
        self.flow_views = None
        self.have_flow = False
        self.unconsumed = 0

    def invalidate_carry(self):
        capi.check(self.lib.dfvo_session_invalidate_carry(self.handle))

    def quiesce(self):
        """a plain solver entry point is about to use the tracker's buffers: wait for the speculative stage and abandon its
        homography half (the keypoints already copied to the host stay valid)"""
        if self.spec_inflight:
            capi.check(self.lib.dfvo_session_quiesce(self.handle))
            self.spec_inflight = False

    # -- DeepModel.forward_depth ---------------------------------------------------------------------------------
    def accepts(self, img):
        return (self.active and self.handle is not None and isinstance(img, np.ndarray) and img.dtype == np.uint8
                and img.shape == (self.h, self.w, 3))

    def push(self, img):
        """upload + enqueue everything frame `img` allows; returns the raw depth [feed_h, feed_w] float32 (read-only view of the
        session's pinned ring, see the module docstring for its lifetime)"""
        a = np.ascontiguousarray(img)
        speculate = self.unconsumed < self.IDLE_PUSHES
        kp = self.kp_cfg if speculate else None
        pose = self.pose_cfg_fn() if (kp is not None and self.pose_cfg_fn is not None) else None
        self._detach(self.gen + 1 - RING)  # the slot this push writes: still referenced -> its holders keep the old buffers
        g = C.c_longlong()
        capi.check(self.lib.dfvo_session_push_frame(self.handle, capi.as_ptr(a), C.byref(kp) if kp is not None else None,
                                                    C.byref(pose) if pose is not None else None,
                                                    0 if speculate else capi.PUSH_NO_FLOW, C.byref(g)))
        self.gen = int(g.value)
        self.kp_spec = kp
        self.have_flow = speculate and self.gen >= 1
        self.spec_inflight = kp is not None and self.gen >= 1
        self.flow_views = None
        self.unconsumed += 1
        self.stats["push"] += 1
        self.stats["push_no_flow"] += 0 if speculate else 1
        p = C.c_void_p()
        capi.check(self.lib.dfvo_session_depth(self.handle, self.gen, C.byref(p)))  # (DfvoError: f16 range exceeded in the depth net)
        return _pinned(p, C.c_float, (self.fh, self.fw), self._lease(self.gen))

    # -- DeepModel.forward_flow ----------------------------------------------------------------------------------
    def _is_frame(self, img, gen):
        """is `img` the frame that was uploaded as generation `gen`?  Compared byte for byte with the session's pinned copy of
        the upload (also when it is the very object that was pushed: it may have been edited in place since)"""
        if gen < 0 or gen < self.gen - 1 or not self.accepts(img):
            return False
        p = C.c_void_p()
        capi.check(self.lib.dfvo_session_frame(self.handle, gen, C.byref(p)))
        return same_bytes(np.ascontiguousarray(img), _pinned(p, C.c_ubyte, (self.h, self.w, 3)))

    def holds_pair(self, ref_img, cur_img):
        self.unconsumed = 0  # the caller does ask for flows: pushes enqueue the flow pass (again)
        return (self.gen >= 1 and self.have_flow and self._is_frame(cur_img, self.gen)
                and self._is_frame(ref_img, self.gen - 1))

    def flow(self):
        """(fwd [2,H,W], bwd [2,H,W], diff [H,W,1]) of (gen - 1, gen): read-only SessionArray views of the pinned host buffers"""
        if self.flow_views is None or self.flow_views[0] != self.gen:
            if self.pose_early:
                self._enqueue_pose_ahead()  # waits for the keypoint count only: the chain starts while the flow is still copied
            pf, pb, pd = C.c_void_p(), C.c_void_p(), C.c_void_p()
            capi.check(self.lib.dfvo_session_flow(self.handle, self.gen, C.byref(pf), C.byref(pb), C.byref(pd)))
            out = []
            lease = self._lease(self.gen)
            for name, p, shape in (("fwd", pf, (2, self.h, self.w)), ("bwd", pb, (2, self.h, self.w)), ("diff", pd, (self.h, self.w, 1))):
                v = _pinned(p, C.c_float, shape, lease).view(SessionArray)
                v._dfvo_tok = (self.sid, self.gen, name)
                out.append(v)
            self.flow_views = (self.gen,) + tuple(out)
            if not self.pose_early:
                self._enqueue_pose_ahead()
        return self.flow_views[1:]

    def _enqueue_pose_ahead(self):
        """the host is back for the first time since the push and the flow exists: the rest of compute_pose_2d2d can run now,
        under np.random's state as it is now (validated again when compute_pose_2d2d is called)"""
        if not (self.pose_ahead and self.kp_spec is not None and self.pose_cfg_fn is not None and self.rng_words is not None):
            return
        cfg, words, enq = self.pose_cfg_fn(), self.rng_words(), C.c_int()
        capi.check(self.lib.dfvo_session_pose_ahead(self.handle, self.gen, capi.as_ptr(words), C.byref(cfg), C.byref(enq)))
        if enq.value:
            self.spec_inflight = True

    # -- KeypointSampler.kp_selection ------------------------------------------------------------------------------
    def _is_buffer(self, arr, name):
        """does `arr` still hold the contents of this generation's buffer `name`?  The token (survives copies, dropped by
        writes through the array object) is only the cheap first test: the WHOLE contents are compared with the pinned buffer
        (~0.3 ms for the flow, behind the device's five-point chain), so a write through a child view, np.copyto, .fill or a
        cv2 call with dst= cannot pass"""
        if getattr(arr, "_dfvo_tok", None) != (self.sid, self.gen, name) or self.flow_views is None or self.flow_views[0] != self.gen:
            return False
        mine = self.flow_views[1 + ("fwd", "bwd", "diff").index(name)]
        return same_bytes(np.asarray(arr), np.asarray(mine))

    def keypoints(self, flow, diff, kp_cfg):
        """local_bestN of this generation if `flow` / `diff` are the session's forward flow / consistency map and the selection
        ran with `kp_cfg`: (kp_ref [n,2], kp_cur [n,2], n, good_kp_found) -- else None"""
        if self.kp_spec is None or bytes(self.kp_spec) != bytes(kp_cfg):
            return None
        if not (self._is_buffer(flow, "fwd") and self._is_buffer(diff, "diff")):
            return None
        pr, pc, n, good = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        capi.check(self.lib.dfvo_session_keypoints(self.handle, self.gen, C.byref(kp_cfg), C.byref(pr), C.byref(pc), C.byref(n),
                                                   C.byref(good)))
        nn = int(n.value)
        if not good.value or nn <= 0:
            return None, None, nn, int(good.value)
        # (owned copies: no view of the keypoint ring leaves the session)
        return np.array(_pinned(pr, C.c_double, (nn, 2))), np.array(_pinned(pc, C.c_double, (nn, 2))), nn, int(good.value)

    # -- EssTracker.compute_pose_2d2d ------------------------------------------------------------------------------
    def pose_2d2d(self, kp_ref, kp_cur, n, cfg, out, inliers, rng_words):
        """rng_words: np.random's state at this call (625 uint32); the advanced state is left on the device"""
        used = C.c_int()
        self.spec_inflight = False  # (the C side waits for the stage -- or, on a mismatch, for its stream -- itself)
        capi.check(self.lib.dfvo_session_pose_2d2d(self.handle, capi.as_ptr(kp_ref), capi.as_ptr(kp_cur), n, C.byref(cfg),
                                                   C.byref(out), capi.as_ptr(inliers), capi.as_ptr(rng_words), C.byref(used)))
        self.stats["pose_resident" if used.value else "pose_plain"] += 1
        if used.value == 2:
            self.stats["pose_ahead"] += 1
        return bool(used.value)
