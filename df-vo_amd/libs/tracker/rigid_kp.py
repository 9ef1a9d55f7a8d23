"""kp_selection_good_depth shared by EssTracker and PnpTracker (/root/reference/libs/tracker/E_tracker.py:645-705 and
pnp_tracker.py:148-213 are the same code): RigidFlow layer + optical-rigid flow distance + opt_rigid_flow_kp on the
device (dfvo_kp_rigid_flow)."""
import ctypes as C

import numpy as np

from ... import capi
from . import _ctx


def kp_selection_good_depth(cfg_all, cam_intrinsics, cur_data, ref_data, rigid_kp_score_method):
    """-> {'kp1_depth','kp2_depth','kp1_depth_uniform','kp2_depth_uniform' [1,N,2], 'rigid_flow_mask' [H,W]}; also stores
    ref_data['rigid_flow_diff'] like the reference"""
    outputs = {}
    if not cfg_all.kp_selection.rigid_flow_kp.enable:
        return outputs
    c = cfg_all.kp_selection.rigid_flow_kp
    h, w = cur_data['depth'].shape
    flow = np.ascontiguousarray(ref_data['flow'], dtype=np.float32)
    diff = np.ascontiguousarray(ref_data['flow_diff'], dtype=np.float32).reshape(h, w)
    raw_depth = np.ascontiguousarray(ref_data['raw_depth'], dtype=np.float32)
    assert flow.shape == (2, h, w) and raw_depth.shape == (h, w)
    K = np.asarray(cam_intrinsics.mat, dtype=np.float64)
    Kinv = np.asarray(cam_intrinsics.inv_mat, dtype=np.float64)
    T = np.asarray(ref_data['rigid_flow_pose'].pose, dtype=np.float64)
    cfg = capi.RigidKpCfg(num_row=int(c.num_row), num_col=int(c.num_col), num_bestN=int(c.num_bestN),
                          rigid_flow_thre=float(c.rigid_flow_thre), optical_flow_thre=float(c.optical_flow_thre),
                          score_method=1 if rigid_kp_score_method == "rigid_flow" else 0)
    for i in range(9):
        cfg.K[i] = K.flat[i]
        cfg.Kinv[i] = Kinv.flat[i]
    for i in range(16):
        cfg.T_ref_to_cur[i] = T.flat[i]
    nmax = int(c.num_bestN)
    kps = [np.zeros((nmax, 2)) for _ in range(4)]
    n = C.c_int()
    rdiff = np.zeros((h, w), np.float32)
    capi.check(capi.lib().dfvo_kp_rigid_flow(_ctx.tracker_exclusive(), capi.as_ptr(flow), capi.as_ptr(diff), capi.as_ptr(raw_depth),
                                             h, w, C.byref(cfg), None, capi.as_ptr(kps[0]), capi.as_ptr(kps[1]),
                                             capi.as_ptr(kps[2]), capi.as_ptr(kps[3]), C.byref(n), capi.as_ptr(rdiff)))
    assert n.value != 0, "sampling threshold is too small."
    ref_data['rigid_flow_diff'] = np.expand_dims(rdiff, 2)
    outputs['kp1_depth'] = kps[0][None, :n.value].copy()
    outputs['kp2_depth'] = kps[1][None, :n.value].copy()
    outputs['kp1_depth_uniform'] = kps[2][None, :n.value].copy()
    outputs['kp2_depth_uniform'] = kps[3][None, :n.value].copy()
    outputs['rigid_flow_mask'] = rdiff
    return outputs
