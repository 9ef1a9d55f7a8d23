"""Host driver of the fused per-pair pipeline (dfvo_pipeline_*): configuration marshalling, the
double-buffered net/solver software pipeline, and pose accumulation as in
/root/reference/libs/dfvo.py:109-119 (update_global_pose) and :121-262 (tracking, hybrid path)."""
import ctypes as C

import numpy as np
import torch

from . import capi

DEFAULTS = dict(  # options/examples/default_configuration.yml
    net_min_depth=0.1, net_max_depth=100.0, baseline_mult=5.4, min_depth=0.0, max_depth=50.0,
    depth_crop=((0.3, 1.0), (0.0, 1.0)), kp_num_row=10, kp_num_col=10, kp_num_bestN=2000, kp_thre=0.1,
    e_reproj_thre=0.2, e_repeat=5, e_max_iters=1000, scale_min_samples=3, scale_max_trials=100,
    scale_stop_prob=0.99, scale_thre=0.1, seed=4869, pnp_repeat=5, pnp_iters=100, pnp_reproj_thre=1.0)

STATUS = {0: "E", 1: "constant_motion", 2: "needs_pnp", 3: "PnP"}


class TrackingPipeline:
    def __init__(self, img_h, img_w, feed_h, feed_w, K, flow_sd, depth_sd, **overrides):
        capi.require_gpu()
        self.lib = capi.lib()
        o = dict(DEFAULTS)
        o.update(overrides)
        self.opts = o
        self.H, self.W, self.feed_h, self.feed_w = img_h, img_w, feed_h, feed_w
        K = np.asarray(K, np.float64)
        self.K = K
        cfg = capi.PipelineCfg(img_h=img_h, img_w=img_w, feed_h=feed_h, feed_w=feed_w,
                               net_min_depth=o["net_min_depth"], net_max_depth=o["net_max_depth"],
                               baseline_mult=o["baseline_mult"], min_depth=o["min_depth"], max_depth=o["max_depth"],
                               fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], kp_num_row=o["kp_num_row"],
                               kp_num_col=o["kp_num_col"], kp_num_bestN=o["kp_num_bestN"], kp_thre=o["kp_thre"],
                               e_reproj_thre=o["e_reproj_thre"], e_repeat=o["e_repeat"], e_max_iters=o["e_max_iters"],
                               scale_min_samples=o["scale_min_samples"], scale_max_trials=o["scale_max_trials"],
                               scale_stop_prob=o["scale_stop_prob"], scale_thre=o["scale_thre"], seed=o["seed"],
                               pnp_repeat=o["pnp_repeat"], pnp_iters=o["pnp_iters"],
                               pnp_reproj_thre=o["pnp_reproj_thre"])
        (y0, y1), (x0, x1) = o["depth_crop"]
        for i, v in enumerate((y0, y1, x0, x1)):
            cfg.depth_crop[i] = v
        KinvT, Kinv = np.linalg.inv(K.T), np.linalg.inv(K)
        for i in range(9):
            cfg.KinvT[i] = KinvT.flat[i]
            cfg.Kinv[i] = Kinv.flat[i]
        h = C.c_void_p()
        capi.check(self.lib.dfvo_pipeline_create(C.byref(cfg), C.byref(h)))
        self.h = h
        nh, nw = self._net_size(img_h, img_w)
        params = {k: (v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, np.float32))
                  for k, v in flow_sd.items()}
        for l in range(1, 7):
            params["aux.linspace_x.%d" % l] = torch.linspace(-1.0, 1.0, nw >> (l - 1)).numpy()
            params["aux.linspace_y.%d" % l] = torch.linspace(-1.0, 1.0, nh >> (l - 1)).numpy()
        capi.set_params(self.lib.dfvo_pipeline_set_flow_param, h, params)
        dparams = {k: v.detach().cpu().float().numpy() for k, v in depth_sd.items()
                   if hasattr(v, "detach") and v.dim() > 0 and "num_batches_tracked" not in k}
        capi.set_params(self.lib.dfvo_pipeline_set_depth_param, h, dparams)
        capi.check(self.lib.dfvo_pipeline_finalize(h))

    @staticmethod
    def _net_size(h, w):
        from .synthetic import _net_size
        return _net_size(h, w)

    def close(self):
        if self.h is not None:
            self.lib.dfvo_pipeline_destroy(self.h)
            self.h = None

    def seed(self, seed):
        capi.check(self.lib.dfvo_pipeline_seed(self.h, int(seed) & 0xffffffff))

    def set_graph(self, enable):
        capi.check(self.lib.dfvo_pipeline_set_graph(self.h, int(enable)))

    def enqueue_nets(self, slot, d_ref, d_cur, d_feed=None):
        """device uint8 tensors (torch.cuda): ref/cur [H,W,3]; feed [feed_h,feed_w,3] = the LANCZOS-resized current
        frame, or None to have the pipeline resize d_cur on the device.
        d_ref=None: the reference frame is the current frame of the previous enqueue_nets call -- the flow net carries that
        frame's image / feature pyramids over instead of running Features on it again (sequences)."""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        capi.check(self.lib.dfvo_pipeline_enqueue_nets(self.h, slot, p(d_ref), p(d_cur), p(d_feed)))

    def prefetch_track(self, slot, flow=None, diff=None):
        """enqueue keypoint selection + the homography chain of `slot` now (call right after enqueue_nets(slot));
        track(slot) then only runs the RandomState consumers"""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        capi.check(self.lib.dfvo_pipeline_prefetch_track(self.h, slot, p(flow), p(diff)))

    def track(self, slot, flow=None, diff=None, depth=None):
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        out = capi.TrackOut()
        capi.check(self.lib.dfvo_pipeline_track(self.h, slot, p(flow), p(diff), p(depth), C.byref(out)))
        return out

    def track_begin(self, slot, flow=None, diff=None, depth=None):
        """first half of track(): enqueues the RandomState-ordered chain of `slot` and returns; the host can feed the nets of
        the pairs ahead while it runs.  Pair k + 1 must not be begun before track_end(k) returned."""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        capi.check(self.lib.dfvo_pipeline_track_begin(self.h, slot, p(flow), p(diff), p(depth)))

    def track_end(self, slot):
        out = capi.TrackOut()
        capi.check(self.lib.dfvo_pipeline_track_end(self.h, slot, C.byref(out)))
        return out

    def set_ref_depth(self, d_feed=None, depth=None):
        """depth of the first reference frame: the uint8 feed image (runs the depth net) or a processed depth map"""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        capi.check(self.lib.dfvo_pipeline_set_ref_depth(self.h, p(d_feed), p(depth)))

    def set_ref_image(self, d_img):
        """depth of the first reference frame from the full-size uint8 frame (device LANCZOS resize + depth net)"""
        capi.check(self.lib.dfvo_pipeline_set_ref_image(self.h, C.c_void_p(d_img.data_ptr())))

    def sync(self):
        capi.check(self.lib.dfvo_pipeline_sync(self.h))

    def net_flops(self):
        return self.lib.dfvo_pipeline_net_flops(self.h)

    def get_outputs(self, slot):
        px = self.H * self.W
        fwd = np.zeros((2, self.H, self.W), np.float32)
        bwd = np.zeros((2, self.H, self.W), np.float32)
        diff = np.zeros((self.H, self.W), np.float32)
        raw = np.zeros((self.H, self.W), np.float32)
        dep = np.zeros((self.H, self.W), np.float64)
        capi.check(self.lib.dfvo_pipeline_get_flow(self.h, slot, capi.as_ptr(fwd), capi.as_ptr(bwd), capi.as_ptr(diff),
                                                   capi.as_ptr(raw), capi.as_ptr(dep)))
        return fwd, bwd, diff, raw, dep

    def get_keypoints(self, slot, cap=4096):
        """kp_best of the reference / current frame [n,2] f64 and the E-tracker's inlier mask [n] bool for `slot`"""
        kr = np.zeros((cap, 2))
        kc = np.zeros((cap, 2))
        inl = np.zeros(cap, np.uint8)
        n = C.c_int(0)
        capi.check(self.lib.dfvo_pipeline_get_keypoints(self.h, slot, cap, capi.as_ptr(kr), capi.as_ptr(kc), capi.as_ptr(inl),
                                                        C.byref(n)))
        assert n.value <= cap
        return kr[:n.value], kc[:n.value], inl[:n.value].astype(bool)

    def get_rng_state(self):
        """the device-resident numpy RandomState as np.random.get_state() would return it"""
        st = np.zeros(625, np.uint32)
        capi.check(self.lib.dfvo_pipeline_get_rng_state(self.h, capi.as_ptr(st)))
        return ("MT19937", st[:624].copy(), int(st[624]), 0, 0.0)

    def set_rng_state(self, state):
        st = np.zeros(625, np.uint32)
        st[:624] = state[1]
        st[624] = state[2]
        capi.check(self.lib.dfvo_pipeline_set_rng_state(self.h, capi.as_ptr(st)))

    # ---- hybrid pose + accumulation (dfvo.py:109-119, 163-262) -------------------------------------------
    @staticmethod
    def hybrid_pose(out, prev_motion):
        """relative pose cur -> ref as a 4x4; `prev_motion` is used for the constant-motion fallback"""
        T = np.eye(4)
        if out.status == 1:
            return prev_motion.copy(), "constant_motion"
        if out.status == 2:
            raise capi.DfvoError("PnP fallback required but no reference depth is known: call set_ref_depth() "
                                 "for the first frame")
        if out.status == 3:  # pnp_tracker.py:112-118: SE3 from (R, t) of solvePnP, then pose = inv_pose
            T[:3, :3] = np.array(out.R[:]).reshape(3, 3)
            T[:3, 3] = np.array(out.t[:])
            return np.linalg.inv(T), "PnP"
        T[:3, :3] = np.array(out.R[:]).reshape(3, 3)
        T[:3, 3] = np.array(out.t[:]) * out.scale
        return T, "E"

    @staticmethod
    def accumulate(global_pose, rel):
        """update_global_pose (dfvo.py:109-119) with scale 1: t_w += R_w t ; R_w = R_w R"""
        g = global_pose.copy()
        g[:3, 3:] = g[:3, :3] @ rel[:3, 3:] + g[:3, 3:]
        g[:3, :3] = g[:3, :3] @ rel[:3, :3]
        return g
