"""EssTracker with the reference's surface (/root/reference/libs/tracker/E_tracker.py:124-307,442-507,
571-643) over the C ABI: every cv2 / sklearn / gric call of the reference runs in HIP kernels
(dfvo_compute_pose_2d2d, dfvo_find_scale_from_depth); this class only marshals numpy arrays and threads
the global np.random stream through the device.  No CPU fallback."""
import copy
import ctypes as C
import os

import numpy as np

from ... import capi
from ..geometry.camera_modules import SE3
from . import _ctx, rigid_kp


# dfvo_tracker_stage_ms slots -> the reference's Timer keys and groups (E_tracker.py:197-296,597-638)
STAGE_KEYS = [("find H", "E-tracker"), ("GRIC-H", "E-tracker"), ("find-Ess", "E-tracker"), ("GRIC-E", "E-tracker"),
              ("find-Ess (full)", "E-tracker"), ("recover pose", "E-tracker"), ("triangulation", "scale_recovery"),
              ("scale ransac", "scale_recovery")]


def feed_timers(timers, slots):
    """append the device times of the last solver call to the reference's Timer under its own sub-keys: the stages run
    as kernels behind one C call here, so their durations come from HIP events between them (dfvo_tracker_stage_ms), not
    from host clocks around Python statements.  `timers` is the reference's libs.general.timer.Timer (or None)."""
    if timers is None and not os.environ.get("DFVO_STAGE_OFFSETS"):
        return
    ms = np.zeros(8)
    capi.check(capi.lib().dfvo_tracker_stage_ms(_ctx.tracker(), capi.as_ptr(ms)))  # (DFVO_STAGE_OFFSETS: prints the marks)
    if timers is None:
        return
    for i in slots:
        if ms[i] < 0:
            continue
        key, group = STAGE_KEYS[i]
        if timers.timers.get(key, -1) == -1:
            timers.add(key, group)
        timers.timers[key]['duration'].append(ms[i] * 1e-3)


VALIDITY_METHODS = {"GRIC": 0, "flow": 1, "homo_ratio": 2}   # include/dfvo_hip.h DFVO_VALIDITY_*
SCALE_METHODS = {"depth_ratio": 0, "abs_diff": 1}             # DFVO_SCALE_*


class EssTracker:
    def __init__(self, cfg, cam_intrinsics, timers):
        self.cfg = cfg
        self.prev_scale = 0
        self.prev_pose = SE3()
        self.cam_intrinsics = cam_intrinsics
        self.timers = timers
        self.max_iters = 1000  # OpenCV 3.4.3's fixed findEssentialMat budget
        # the frame session enqueues the RandomState-independent half of compute_pose_2d2d behind the flow net with this
        # configuration (libs/deep_models/session.py); registered only where that half does not depend on the call's arguments
        if self.cfg.e_tracker.validity.method in VALIDITY_METHODS:
            _ctx.register_pose_cfg(lambda: self._pose_cfg(True))

    def _pose_cfg(self, is_iterative):
        valid_cfg = self.cfg.e_tracker.validity
        K = np.asarray(self.cam_intrinsics.mat, dtype=np.float64)
        repeat = int(self.cfg.e_tracker.ransac.repeat) if is_iterative else 3
        cfg = capi.Pose2d2dCfg(fx=float(self.cam_intrinsics.fx), cx=float(self.cam_intrinsics.cx),
                               cy=float(self.cam_intrinsics.cy),
                               reproj_thre=float(self.cfg.e_tracker.ransac.reproj_thre), repeat=repeat,
                               max_iters=self.max_iters, validity_method=VALIDITY_METHODS[valid_cfg.method],
                               validity_thre=0.0 if valid_cfg.method == "GRIC" else float(valid_cfg.thre))
        KinvT, Kinv = np.linalg.inv(K.T), np.linalg.inv(K)
        for i in range(9):
            cfg.KinvT[i] = KinvT.flat[i]
            cfg.Kinv[i] = Kinv.flat[i]
        return cfg

    def compute_pose_2d2d(self, kp_ref, kp_cur, is_iterative):
        """E_tracker.py:154-307 -> {'pose': SE3 (cur -> ref), 'inliers': bool [N]}"""
        valid_cfg = self.cfg.e_tracker.validity
        if valid_cfg.method not in VALIDITY_METHODS:
            raise NotImplementedError("e_tracker.validity.method '%s' (the reference knows GRIC, flow, homo_ratio)"
                                      % valid_cfg.method)
        kp_ref = np.ascontiguousarray(kp_ref, dtype=np.float64)
        kp_cur = np.ascontiguousarray(kp_cur, dtype=np.float64)
        n = kp_ref.shape[0]
        cfg = self._pose_cfg(is_iterative)
        out = capi.Pose2d2dOut()
        inl = np.zeros(max(n, 1), np.uint8)
        if _ctx.session is not None:
            # (the C side compares keypoints, configuration and RandomState with what the device holds: on a full match the
            # call was enqueued ahead and only its result is fetched; with matching keypoints only the RandomState-consuming
            # half is left to run; else the plain entry point -- always under np.random's state as it is now)
            _ctx.session.pose_2d2d(kp_ref, kp_cur, n, cfg, out, inl, _ctx.numpy_rng_words())
        else:
            _ctx.push_numpy_rng()
            capi.check(capi.lib().dfvo_compute_pose_2d2d(_ctx.tracker_exclusive(), capi.as_ptr(kp_ref), capi.as_ptr(kp_cur), n,
                                                         C.byref(cfg), C.byref(out), capi.as_ptr(inl)))
        _ctx.pull_numpy_rng()
        feed_timers(self.timers, range(0, 6))
        pose = SE3()
        pose.R = np.array(out.R[:]).reshape(3, 3)
        pose.t = np.array(out.t[:]).reshape(3, 1)
        self.last_diag = out
        return {"pose": pose, "inliers": inl[:n] == 1}

    def scale_recovery(self, cur_data, ref_data, E_pose, is_iterative):
        """E_tracker.py:442-471"""
        outputs = {}
        if self.cfg.scale_recovery.method == "simple":
            outputs['scale'] = self.scale_recovery_simple(cur_data, ref_data, E_pose, is_iterative)
        elif self.cfg.scale_recovery.method == "iterative":
            iter_outputs = self.scale_recovery_iterative(cur_data, ref_data, E_pose)
            outputs['scale'] = iter_outputs['scale']
            outputs['cur_kp_depth'] = iter_outputs['cur_kp']
            outputs['ref_kp_depth'] = iter_outputs['ref_kp']
            outputs['rigid_flow_mask'] = iter_outputs['rigid_flow_mask']
        else:
            raise NotImplementedError("scale_recovery.method '%s'" % self.cfg.scale_recovery.method)
        return outputs

    def scale_recovery_iterative(self, cur_data, ref_data, E_pose):
        """E_tracker.py:509-569: up to five rounds of (rigid-flow keypoints under the current scale -> depth-ratio scale)"""
        outputs = {}
        scale = self.prev_scale
        delta = 0.001
        for _ in range(5):
            rigid_flow_pose = copy.deepcopy(E_pose)
            rigid_flow_pose.t *= scale
            ref_data['rigid_flow_pose'] = SE3(rigid_flow_pose.inv_pose)
            kp_sel_outputs = self.kp_selection_good_depth(cur_data, ref_data,
                                                          self.cfg.scale_recovery.iterative_kp.score_method)
            ref_data['kp_depth'] = kp_sel_outputs['kp1_depth_uniform'][0]
            cur_data['kp_depth'] = kp_sel_outputs['kp2_depth_uniform'][0]
            cur_data['rigid_flow_mask'] = kp_sel_outputs['rigid_flow_mask']
            cur_kp = cur_data[self.cfg.scale_recovery.kp_src]
            ref_kp = ref_data[self.cfg.scale_recovery.kp_src]
            new_scale = self.find_scale_from_depth(ref_kp, cur_kp, E_pose.inv_pose, cur_data['depth'])
            delta_scale = np.abs(new_scale - scale)
            scale = new_scale
            self.prev_scale = new_scale
            outputs['scale'] = scale
            outputs['cur_kp'] = cur_data['kp_depth']
            outputs['ref_kp'] = ref_data['kp_depth']
            outputs['rigid_flow_mask'] = cur_data['rigid_flow_mask']
            if delta_scale < delta:
                return outputs
        return outputs

    def kp_selection_good_depth(self, cur_data, ref_data, rigid_kp_score_method):
        """E_tracker.py:645-705 (see rigid_kp.kp_selection_good_depth)"""
        return rigid_kp.kp_selection_good_depth(self.cfg, self.cam_intrinsics, cur_data, ref_data, rigid_kp_score_method)

    def scale_recovery_simple(self, cur_data, ref_data, E_pose, is_iterative):
        """E_tracker.py:473-507"""
        src = self.cfg.scale_recovery.iterative_kp.kp_src if is_iterative else self.cfg.scale_recovery.kp_src
        return self.find_scale_from_depth(ref_data[src], cur_data[src], E_pose.inv_pose, cur_data['depth'])

    def find_scale_from_depth(self, kp1, kp2, T_21, depth2):
        """E_tracker.py:571-643 (ransac.method 'depth_ratio' or 'abs_diff')"""
        rc = self.cfg.scale_recovery.ransac
        if rc.method not in SCALE_METHODS:
            raise NotImplementedError("scale_recovery.ransac.method '%s'" % rc.method)
        kp1 = np.ascontiguousarray(kp1, dtype=np.float64)
        kp2 = np.ascontiguousarray(kp2, dtype=np.float64)
        T_21 = np.ascontiguousarray(T_21, dtype=np.float64)
        depth2 = np.asarray(depth2)
        h, w = depth2.shape[:2]
        # depth2 is only read under the sparse triangulated map (E_tracker.py:604-612, ops_3d.py:29-40), i.e. at the truncated
        # kp2 pixels: those n values travel instead of the H x W float64 map (3.7 MB per pair at KITTI size)
        tx, ty = np.trunc(kp2[:, 0]), np.trunc(kp2[:, 1])
        with np.errstate(invalid="ignore"):
            inside = (tx >= 0) & (tx < w) & (ty >= 0) & (ty < h)  # (NaN / inf coordinates compare False)
        depth_at_kp2 = np.zeros(kp2.shape[0], np.float64)
        depth_at_kp2[inside] = depth2.reshape(h, w)[ty[inside].astype(np.int64), tx[inside].astype(np.int64)]
        cam = self.cam_intrinsics
        scfg = capi.ScaleCfg(cx=float(cam.cx), cy=float(cam.cy), fx=float(cam.fx), fy=float(cam.fy),
                             min_samples=int(rc.min_samples), max_trials=int(rc.max_trials),
                             stop_prob=float(rc.stop_prob), thre=float(rc.thre), method=SCALE_METHODS[rc.method])
        scale = C.c_double()
        info = np.zeros(4, np.int32)
        rng = _ctx.numpy_rng_words()  # in: np.random's state; out: the state after sklearn's draws
        capi.check(capi.lib().dfvo_find_scale_from_depth_at_kp(_ctx.tracker_exclusive(), capi.as_ptr(kp1), capi.as_ptr(kp2),
                                                               kp1.shape[0], capi.as_ptr(T_21), capi.as_ptr(depth_at_kp2), h, w,
                                                               C.byref(scfg), capi.as_ptr(rng), C.byref(scale), capi.as_ptr(info)))
        _ctx.set_numpy_rng(rng)
        feed_timers(self.timers, (6, 7))
        if info[3] < 0:
            raise ValueError("RANSAC could not find a valid consensus set (sklearn RANSACRegressor semantics)")
        return -1 if scale.value == -1.0 else scale.value

    def compute_rigid_flow_kp(self, cur_data, ref_data, pose):
        """E_tracker.py:422-440"""
        rigid_pose = copy.deepcopy(pose)
        ref_data['rigid_flow_pose'] = SE3(rigid_pose.inv_pose)
        kp_sel_outputs = self.kp_selection_good_depth(cur_data, ref_data, self.cfg.e_tracker.iterative_kp.score_method)
        ref_data['kp_depth'] = kp_sel_outputs['kp1_depth'][0]
        cur_data['kp_depth'] = kp_sel_outputs['kp2_depth'][0]
        ref_data['kp_depth_uniform'] = kp_sel_outputs['kp1_depth_uniform'][0]
        cur_data['kp_depth_uniform'] = kp_sel_outputs['kp2_depth_uniform'][0]
        cur_data['rigid_flow_mask'] = kp_sel_outputs['rigid_flow_mask']
