"""PnpTracker with the reference's surface (/root/reference/libs/tracker/pnp_tracker.py:22-125).
The fallback path (E-tracker rejected or scale == -1, dfvo.py:225-250)."""
import copy

import numpy as np

from ... import capi
from ..geometry.camera_modules import SE3
from . import _ctx, rigid_kp


class PnpTracker:
    def __init__(self, cfg, cam_intrinsics):
        self.cfg = cfg
        self.cam_intrinsics = cam_intrinsics

    def compute_pose_3d2d(self, kp1, kp2, depth_1, is_iterative):
        """pnp_tracker.py:45-125 -> {'pose': SE3 (view-2 -> view-1), 'kp1', 'kp2'}"""
        import ctypes as C
        lib = capi.lib()
        kp1 = np.ascontiguousarray(kp1, dtype=np.float64)
        kp2 = np.ascontiguousarray(kp2, dtype=np.float64)
        depth_1 = np.asarray(depth_1)
        h, w = depth_1.shape
        n = kp1.shape[0]
        # depth_1 is only read at kp1's pixels (pnp_tracker.py:71-77: astype(int) truncation, negative indices wrap): those n
        # values travel instead of the H x W float64 map
        xi, yi = np.trunc(kp1[:, 0]), np.trunc(kp1[:, 1])
        with np.errstate(invalid="ignore"):
            xi, yi = np.where(xi < 0, xi + w, xi), np.where(yi < 0, yi + h, yi)
            inside = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)  # (NaN / inf compare False)
        depth_at_kp1 = np.zeros(n, np.float64)
        depth_at_kp1[inside] = depth_1[yi[inside].astype(np.int64), xi[inside].astype(np.int64)]
        cam = self.cam_intrinsics
        repeat = int(self.cfg.pnp_tracker.ransac.repeat) if is_iterative else 3
        cfg = capi.Pose3d2dCfg(fx=float(cam.fx), fy=float(cam.fy), cx=float(cam.cx), cy=float(cam.cy),
                               min_depth=float(self.cfg.depth.min_depth), max_depth=float(self.cfg.depth.max_depth),
                               iters=int(self.cfg.pnp_tracker.ransac.iter),
                               reproj_thre=float(self.cfg.pnp_tracker.ransac.reproj_thre), repeat=repeat)
        Kinv = np.asarray(cam.inv_mat, dtype=np.float64)
        for i in range(9):
            cfg.Kinv[i] = Kinv.flat[i]
        out = capi.Pose3d2dOut()
        keep = np.zeros(max(n, 1), np.uint8)
        rng = _ctx.numpy_rng_words()  # in: np.random's state; out: the state after the shuffles
        capi.check(lib.dfvo_compute_pose_3d2d_at_kp(_ctx.tracker_exclusive(), capi.as_ptr(kp1), capi.as_ptr(kp2), n,
                                                    capi.as_ptr(depth_at_kp1), h, w, C.byref(cfg), capi.as_ptr(rng), C.byref(out),
                                                    capi.as_ptr(keep)))
        _ctx.set_numpy_rng(rng)
        # format pose (pnp_tracker.py:112-118): identity when no repeat produced a model, then inverted
        pose = SE3()
        if out.found:
            pose.R = np.array(out.R[:]).reshape(3, 3)
            pose.t = np.array(out.tvec[:]).reshape(3, 1)
        pose.pose = pose.inv_pose
        sel = keep[:n] == 1
        return {"pose": pose, "kp1": kp1[sel], "kp2": kp2[sel]}

    def compute_rigid_flow_kp(self, cur_data, ref_data, pose):
        """pnp_tracker.py:127-145"""
        rigid_pose = copy.deepcopy(pose)
        ref_data['rigid_flow_pose'] = SE3(rigid_pose.inv_pose)
        kp_sel_outputs = self.kp_selection_good_depth(cur_data, ref_data, self.cfg.pnp_tracker.iterative_kp.score_method)
        ref_data['kp_depth'] = kp_sel_outputs['kp1_depth'][0]
        cur_data['kp_depth'] = kp_sel_outputs['kp2_depth'][0]
        ref_data['kp_depth_uniform'] = kp_sel_outputs['kp1_depth_uniform'][0]
        cur_data['kp_depth_uniform'] = kp_sel_outputs['kp2_depth_uniform'][0]
        cur_data['rigid_flow_mask'] = kp_sel_outputs['rigid_flow_mask']

    def kp_selection_good_depth(self, cur_data, ref_data, rigid_kp_score_method):
        """pnp_tracker.py:148-213 (see rigid_kp.kp_selection_good_depth)"""
        return rigid_kp.kp_selection_good_depth(self.cfg, self.cam_intrinsics, cur_data, ref_data, rigid_kp_score_method)
