"""Frame session under the drop-in classes: the reference's own synchronous call order (/root/reference/libs/dfvo.py:299-345,
121-262), fast, with every call returning exactly what it returned before (C side: df-vo_amd/csrc/session.hip).

DeepModel.forward_depth(k) PUSHES frame k: one upload, the depth net of k and the flow net of (k - 1, k) on their own
streams, and -- once KeypointSampler / EssTracker have registered their configurations (libs/tracker/_ctx.py) -- the keypoint
selection and the RandomState-independent half of compute_pose_2d2d behind the flow net.  The later calls of the frame loop
wait for an event and hand out the result, IF they are asked for what was enqueued:
  * forward_flow checks that it is given the two frames that were pushed (object identity + sampled contents),
  * the flow / consistency arrays it returns are SessionArray views of pinned host buffers; a copy of one (dfvo.py:330-333
    copies them) keeps the generation token, any write through the array drops it, and kp_selection additionally compares a
    strided sample of the contents with the session's buffer before it trusts the device-resident copy,
  * compute_pose_2d2d compares keypoints and configuration byte for byte on the C side; its RandomState-consuming half is
    enqueued AHEAD, from forward_flow, under np.random's state at that moment, and its result is handed out only if
    np.random's state at the compute_pose_2d2d call is still that state, word for word (DFVO_SESSION_POSE_AHEAD=0: never ahead).
Anything else takes the plain host-array entry point -- same results, the round-3 speed."""
import ctypes as C
import os

import numpy as np

from ... import capi

RING = 3  # host buffer sets of the C side: a returned view is valid until two further frames have been pushed


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


def _pinned(ptr, ctype, shape):
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape)
    a.flags.writeable = False
    return a


def _sample_idx(n, k=1024):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


class FrameSession:
    _next_id = 0

    def __init__(self, flow_model, depth_model, tracker, height, width, feed_height, feed_width):
        self.h, self.w, self.fh, self.fw = int(height), int(width), int(feed_height), int(feed_width)
        self.lib = capi.lib()
        self._keep = (flow_model, depth_model, tracker)
        h = C.c_void_p()
        capi.check(self.lib.dfvo_session_create(flow_model, depth_model, tracker, self.h, self.w, C.byref(h)))
        self.handle = h
        FrameSession._next_id += 1
        self.sid = FrameSession._next_id
        self.gen = -1
        self.imgs = {}          # generation -> (the pushed array object, its sampled bytes)
        self.flow_views = None  # (generation, fwd, bwd, diff [H,W,1]) of the newest pair
        self.kp_cfg = None      # capi.SessionKpCfg registered by KeypointSampler (None: no speculative selection)
        self.pose_cfg_fn = None  # () -> capi.Pose2d2dCfg registered by EssTracker
        self.kp_spec = None     # the kp cfg the newest push ran with
        self.spec_inflight = False  # a speculative keypoint / homography stage may be running on the tracker's buffers
        self._img_idx = _sample_idx(self.h * self.w * 3)
        self._flow_idx = {n: _sample_idx(sz) for n, sz in (("fwd", 2 * self.h * self.w), ("bwd", 2 * self.h * self.w),
                                                           ("diff", self.h * self.w))}
        self.pose_ahead = os.environ.get("DFVO_SESSION_POSE_AHEAD", "1") != "0"
        self.rng_words = None   # () -> np.random's state as 625 uint32 words (libs/tracker/_ctx.numpy_rng_words)
        self.stats = {"push": 0, "flow_resident": 0, "flow_plain": 0, "kp_resident": 0, "kp_plain": 0, "pose_resident": 0,
                      "pose_plain": 0, "pose_ahead": 0}

    def close(self):
        if self.handle is not None:
            self.lib.dfvo_session_destroy(self.handle)
            self.handle = None

    def reset(self):
        capi.check(self.lib.dfvo_session_reset(self.handle))
        self.gen = -1
        self.imgs.clear()
        self.flow_views = None

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
        return isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.shape == (self.h, self.w, 3)

    def push(self, img):
        """upload + enqueue everything frame `img` allows; returns the raw depth [feed_h, feed_w] float32 (pinned view)"""
        a = np.ascontiguousarray(img)
        kp = self.kp_cfg
        pose = self.pose_cfg_fn() if (kp is not None and self.pose_cfg_fn is not None) else None
        g = C.c_longlong()
        capi.check(self.lib.dfvo_session_push_frame(self.handle, capi.as_ptr(a), C.byref(kp) if kp is not None else None,
                                                    C.byref(pose) if pose is not None else None, C.byref(g)))
        self.gen = int(g.value)
        self.kp_spec = kp
        self.spec_inflight = kp is not None and self.gen >= 1
        self.imgs[self.gen] = (img, a.reshape(-1)[self._img_idx].copy())
        self.imgs.pop(self.gen - 2, None)
        self.flow_views = None
        self.stats["push"] += 1
        p = C.c_void_p()
        capi.check(self.lib.dfvo_session_depth(self.handle, self.gen, C.byref(p)))
        return _pinned(p, C.c_float, (self.fh, self.fw))

    # -- DeepModel.forward_flow ----------------------------------------------------------------------------------
    def _is_frame(self, img, gen):
        held = self.imgs.get(gen)
        if held is None or not self.accepts(img):
            return False
        if img is not held[0] and not np.array_equal(img, held[0]):
            return False
        return bool(np.array_equal(np.ascontiguousarray(img).reshape(-1)[self._img_idx], held[1]))  # (guards in-place edits)

    def holds_pair(self, ref_img, cur_img):
        return self.gen >= 1 and self._is_frame(cur_img, self.gen) and self._is_frame(ref_img, self.gen - 1)

    def flow(self):
        """(fwd [2,H,W], bwd [2,H,W], diff [H,W,1]) of (gen - 1, gen): SessionArray views of the pinned host buffers"""
        if self.flow_views is None or self.flow_views[0] != self.gen:
            pf, pb, pd = C.c_void_p(), C.c_void_p(), C.c_void_p()
            capi.check(self.lib.dfvo_session_flow(self.handle, self.gen, C.byref(pf), C.byref(pb), C.byref(pd)))
            out = []
            for name, p, shape in (("fwd", pf, (2, self.h, self.w)), ("bwd", pb, (2, self.h, self.w)), ("diff", pd, (self.h, self.w, 1))):
                v = _pinned(p, C.c_float, shape).view(SessionArray)
                v._dfvo_tok = (self.sid, self.gen, name)
                out.append(v)
            self.flow_views = (self.gen,) + tuple(out)
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
        """does `arr` still hold the contents of this generation's buffer `name`? token (survives copies, dropped by writes)
        plus a strided sample of the values against the pinned buffer"""
        if getattr(arr, "_dfvo_tok", None) != (self.sid, self.gen, name) or self.flow_views is None or self.flow_views[0] != self.gen:
            return False
        mine = self.flow_views[1 + ("fwd", "bwd", "diff").index(name)]
        if arr.shape != mine.shape or arr.dtype != mine.dtype:
            return False
        idx = self._flow_idx[name]
        return bool(np.array_equal(np.asarray(arr).reshape(-1)[idx], np.asarray(mine).reshape(-1)[idx]))

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
        return _pinned(pr, C.c_double, (nn, 2)), _pinned(pc, C.c_double, (nn, 2)), nn, int(good.value)

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
