"""ctypes binding of libdfvo_hip.so -- one Python function table, no compute here.

The product path has NO CPU fallback: if the shared library is missing or a call fails, a
``DfvoError`` is raised (the -m gpu tests and the driver rely on that).
"""
import ctypes as C
import os

import numpy as np

from . import LIB_PATH


PUSH_NO_FLOW = 1   # dfvo_session_push_frame flags (include/dfvo_hip.h)
ERR_RANGE = -4     # DFVO_ERR_RANGE: an activation left f16's range under an f16x3 / f16 packing


class DfvoError(RuntimeError):
    pass


_lib = None


class Pose2d2dCfg(C.Structure):
    _fields_ = [("fx", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("reproj_thre", C.c_double),
                ("repeat", C.c_int), ("max_iters", C.c_int), ("KinvT", C.c_double * 9), ("Kinv", C.c_double * 9),
                ("validity_method", C.c_int), ("validity_thre", C.c_double)]


class RigidKpCfg(C.Structure):
    _fields_ = [("num_row", C.c_int), ("num_col", C.c_int), ("num_bestN", C.c_int), ("rigid_flow_thre", C.c_double),
                ("optical_flow_thre", C.c_double), ("score_method", C.c_int), ("K", C.c_double * 9),
                ("Kinv", C.c_double * 9), ("T_ref_to_cur", C.c_double * 16)]


class Pose2d2dOut(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("n", C.c_int), ("best_inlier_cnt", C.c_int),
                ("num_valid", C.c_int), ("major_valid", C.c_int), ("cheirality", C.c_int), ("h_found", C.c_int),
                ("h_gric", C.c_double), ("rep_inliers", C.c_int * 8), ("rep_valid", C.c_int * 8),
                ("rep_gric", C.c_double * 8)]


class ScaleCfg(C.Structure):
    _fields_ = [("cx", C.c_double), ("cy", C.c_double), ("fx", C.c_double), ("fy", C.c_double),
                ("min_samples", C.c_int), ("max_trials", C.c_int), ("stop_prob", C.c_double), ("thre", C.c_double),
                ("method", C.c_int)]


class Pose3d2dCfg(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("Kinv", C.c_double * 9), ("min_depth", C.c_double), ("max_depth", C.c_double),
                ("repeat", C.c_int), ("iters", C.c_int), ("reproj_thre", C.c_double)]


class Pose3d2dOut(C.Structure):
    _fields_ = [("found", C.c_int), ("best_inliers", C.c_int), ("n_filtered", C.c_int), ("status", C.c_int),
                ("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("R", C.c_double * 9)]


class PipelineCfg(C.Structure):
    _fields_ = [("img_h", C.c_int), ("img_w", C.c_int), ("feed_h", C.c_int), ("feed_w", C.c_int),
                ("net_min_depth", C.c_float), ("net_max_depth", C.c_float), ("baseline_mult", C.c_float),
                ("min_depth", C.c_double), ("max_depth", C.c_double), ("depth_crop", C.c_double * 4),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("KinvT", C.c_double * 9), ("Kinv", C.c_double * 9),
                ("kp_num_row", C.c_int), ("kp_num_col", C.c_int), ("kp_num_bestN", C.c_int), ("kp_thre", C.c_double),
                ("e_reproj_thre", C.c_double), ("e_repeat", C.c_int), ("e_max_iters", C.c_int),
                ("scale_min_samples", C.c_int), ("scale_max_trials", C.c_int), ("scale_stop_prob", C.c_double),
                ("scale_thre", C.c_double), ("seed", C.c_uint32), ("pnp_repeat", C.c_int), ("pnp_iters", C.c_int),
                ("pnp_reproj_thre", C.c_double)]


class TrackOut(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("scale", C.c_double), ("status", C.c_int),
                ("n_kp", C.c_int), ("good_kp_found", C.c_int), ("best_inlier_cnt", C.c_int), ("num_valid", C.c_int),
                ("cheirality", C.c_int), ("scale_n_valid", C.c_int), ("scale_n_trials", C.c_int),
                ("scale_n_inliers", C.c_int), ("pnp_found", C.c_int), ("pnp_inliers", C.c_int),
                ("pnp_n_filtered", C.c_int)]


class SessionKpCfg(C.Structure):
    _fields_ = [("num_row", C.c_int), ("num_col", C.c_int), ("num_bestN", C.c_int), ("thre", C.c_float),
                ("score_method", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "N", "H", "W", "kh", "kw", "stride", "pad_h", "pad_w", "pad_mode",
        "c0", "cs0", "co0", "up0", "c1", "cs1", "co1", "cout", "act")] + [
        ("act_param", C.c_float)] + [(n, C.c_int) for n in ("res_cs", "res_co", "dst_cs", "dst_co")]


_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol declared in include/dfvo_hip.h must be listed here
SIGNATURES = {
    "dfvo_last_error": (C.c_char_p, []),
    "dfvo_device_count": (_i, []),
    "dfvo_set_device": (_i, [_i]),
    "dfvo_sync_device": (_i, []),
    "dfvo_malloc": (_i, [C.POINTER(_vp), _sz]),
    "dfvo_free": (_i, [_vp]),
    "dfvo_memcpy_h2d": (_i, [_vp, _vp, _sz]),
    "dfvo_memcpy_d2h": (_i, [_vp, _vp, _sz]),
    "dfvo_conv2d": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dfvo_compose_trajectory": (_i, [_vp, _i, _vp, _vp, C.POINTER(_i)]),
    "dfvo_compose_trajectory_device": (_i, [_vp, _i, _vp, _vp, C.POINTER(_i), _vp]),
    "dfvo_tracker_stage_ms": (_i, [_vp, _vp]),
    "dfvo_comm_unique_id": (_i, [_vp]),
    "dfvo_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "dfvo_comm_destroy": (_i, [_vp]),
    "dfvo_allgather_poses": (_i, [_vp, _vp, _i, _vp, _vp]),
    "dfvo_allgather_poses_device": (_i, [_vp, _vp, _i, _vp, _vp]),
    "dfvo_session_create": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(_vp)]),
    "dfvo_session_destroy": (None, [_vp]),
    "dfvo_session_reset": (_i, [_vp]),
    "dfvo_session_invalidate_carry": (_i, [_vp]),
    "dfvo_session_quiesce": (_i, [_vp]),
    "dfvo_session_push_frame": (_i, [_vp, _vp, _vp, _vp, _i, C.POINTER(C.c_longlong)]),
    "dfvo_session_frame": (_i, [_vp, C.c_longlong, C.POINTER(_vp)]),
    "dfvo_session_detach_slot": (_i, [_vp, C.c_longlong, C.POINTER(_vp)]),
    "dfvo_host_free": (_i, [_vp]),
    "dfvo_session_depth": (_i, [_vp, C.c_longlong, C.POINTER(_vp)]),
    "dfvo_session_flow": (_i, [_vp, C.c_longlong, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "dfvo_session_keypoints": (_i, [_vp, C.c_longlong, _vp, C.POINTER(_vp), C.POINTER(_vp), _ip, _ip]),
    "dfvo_session_pose_ahead": (_i, [_vp, C.c_longlong, _vp, _vp, _ip]),
    "dfvo_session_pose_2d2d": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _ip]),
    "dfvo_set_conv_precision": (_i, [C.c_char_p]),
    "dfvo_get_conv_precision": (C.c_char_p, []),
    "dfvo_set_sklearn_compat": (_i, [C.c_char_p]),
    "dfvo_f16s_overflow_count": (_i, [C.POINTER(C.c_ulonglong), _i]),
    "dfvo_conv_profile_begin": (_i, []),
    "dfvo_conv_profile_end": (_i, [_vp, _vp, _vp]),
    "dfvo_conv_profile_end_bytes": (_i, [_vp, _vp, _vp, _vp]),
    "dfvo_correlation": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "dfvo_backward_warp": (_i, [_vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dfvo_deconv_dw4x4s2": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "dfvo_lanczos_coeffs": (_i, [_i, _i, _vp, _vp, _i, _ip]),
    "dfvo_resize_lanczos_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _vp]),
    "dfvo_resize_linear_u8": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "dfvo_read_image_tail_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "dfvo_resize_bilinear": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "dfvo_flownet_create": (_i, [_i, _i, _vp, C.POINTER(_vp)]),
    "dfvo_flownet_destroy": (None, [_vp]),
    "dfvo_flownet_set_param": (_i, [_vp, C.c_char_p, _vp, _i, _ip]),
    "dfvo_flownet_finalize": (_i, [_vp]),
    "dfvo_flow_target_size": (_i, [_i, _i, _ip, _ip]),
    "dfvo_flownet_net_size": (_i, [_vp, _ip, _ip]),
    "dfvo_flownet_set_graph": (_i, [_vp, _i]),
    "dfvo_flownet_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dfvo_flownet_forward_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dfvo_flownet_last_flops": (_d, [_vp]),
    "dfvo_flownet_get_level_flow": (_i, [_vp, _i, _vp, _ip, _ip]),
    "dfvo_flownet_sync": (_i, [_vp]),
    "dfvo_depthnet_create": (_i, [_i, _i, _f, _f, _f, _vp, C.POINTER(_vp)]),
    "dfvo_depthnet_destroy": (None, [_vp]),
    "dfvo_depthnet_set_param": (_i, [_vp, C.c_char_p, _vp, _i, _ip]),
    "dfvo_depthnet_finalize": (_i, [_vp]),
    "dfvo_depthnet_set_graph": (_i, [_vp, _i]),
    "dfvo_depthnet_forward": (_i, [_vp, _vp, _vp]),
    "dfvo_depthnet_forward_host": (_i, [_vp, _vp, _vp]),
    "dfvo_depthnet_forward_image_host": (_i, [_vp, _vp, _i, _i, _vp]),
    "dfvo_depthnet_last_flops": (_d, [_vp]),
    "dfvo_depthnet_sync": (_i, [_vp]),
    "dfvo_depth_postprocess": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "dfvo_tracker_create": (_i, [_vp, C.POINTER(_vp)]),
    "dfvo_tracker_destroy": (None, [_vp]),
    "dfvo_find_essential_mat": (_i, [_vp, _vp, _vp, _i, _d, _d, _d, _d, _d, _i, _vp, _vp, _vp]),
    "dfvo_find_homography": (_i, [_vp, _vp, _vp, _i, _d, _i, _d, _vp, _vp, _vp]),
    "dfvo_recover_pose": (_i, [_vp, _vp, _vp, _vp, _i, _d, _d, _d, _vp, _vp, _vp, _ip]),
    "dfvo_triangulate_points": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dfvo_tracker_seed": (_i, [_vp, C.c_uint32]),
    "dfvo_tracker_set_rng_state": (_i, [_vp, _vp]),
    "dfvo_tracker_get_rng_state": (_i, [_vp, _vp]),
    "dfvo_kp_local_bestn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _ip, _ip]),
    "dfvo_kp_local_bestn_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _ip, _ip]),
    "dfvo_kp_sampled": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "dfvo_kp_bestn": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "dfvo_kp_rigid_flow": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.POINTER(RigidKpCfg), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dfvo_compute_pose_2d2d": (_i, [_vp, _vp, _vp, _i, C.POINTER(Pose2d2dCfg), C.POINTER(Pose2d2dOut), _vp]),
    "dfvo_pipeline_create": (_i, [C.POINTER(PipelineCfg), C.POINTER(_vp)]),
    "dfvo_pipeline_destroy": (None, [_vp]),
    "dfvo_pipeline_set_flow_param": (_i, [_vp, C.c_char_p, _vp, _i, _ip]),
    "dfvo_pipeline_set_depth_param": (_i, [_vp, C.c_char_p, _vp, _i, _ip]),
    "dfvo_pipeline_finalize": (_i, [_vp]),
    "dfvo_pipeline_seed": (_i, [_vp, C.c_uint32]),
    "dfvo_pipeline_set_graph": (_i, [_vp, _i]),
    "dfvo_pipeline_enqueue_nets": (_i, [_vp, _i, _vp, _vp, _vp]),
    "dfvo_pipeline_track": (_i, [_vp, _i, _vp, _vp, _vp, C.POINTER(TrackOut)]),
    "dfvo_pipeline_track_begin": (_i, [_vp, _i, _vp, _vp, _vp]),
    "dfvo_pipeline_track_end": (_i, [_vp, _i, C.POINTER(TrackOut)]),
    "dfvo_pipeline_set_ref_depth": (_i, [_vp, _vp, _vp]),
    "dfvo_pipeline_set_ref_image": (_i, [_vp, _vp]),
    "dfvo_pipeline_prefetch_track": (_i, [_vp, _i, _vp, _vp]),
    "dfvo_pipeline_get_flow": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "dfvo_pipeline_get_keypoints": (_i, [_vp, _i, _i, _vp, _vp, _vp, _ip]),
    "dfvo_pipeline_get_rng_state": (_i, [_vp, _vp]),
    "dfvo_pipeline_set_rng_state": (_i, [_vp, _vp]),
    "dfvo_pipeline_sync": (_i, [_vp]),
    "dfvo_pipeline_net_flops": (_d, [_vp]),
    "dfvo_find_scale_from_depth": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, C.POINTER(ScaleCfg), C.POINTER(_d), _vp]),
    "dfvo_find_scale_from_depth_at_kp": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, C.POINTER(ScaleCfg), _vp, C.POINTER(_d), _vp]),
    "dfvo_compute_pose_3d2d": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, C.POINTER(Pose3d2dCfg), C.POINTER(Pose3d2dOut), _vp]),
    "dfvo_compute_pose_3d2d_at_kp": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, C.POINTER(Pose3d2dCfg), _vp, C.POINTER(Pose3d2dOut), _vp]),
    "dfvo_ransac_regressor": (_i, [_vp, _vp, _vp, _i, C.POINTER(ScaleCfg), C.POINTER(_d), _vp]),
}


def lib():
    """Load (once) and return the ctypes library; raises DfvoError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DfvoError(
            "libdfvo_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback." % LIB_PATH)
    try:
        # torch bundles its own libamdhip64; import it FIRST so that one HIP runtime serves both
        # (loading ours first made torch.cuda unavailable on the GPU box)
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DfvoError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            raise DfvoError("libdfvo_hip.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().dfvo_last_error()
        raise DfvoError("libdfvo_hip call failed (%d): %s" % (rc, msg.decode() if msg else "?"))


def check_f16_range(seen, what):
    """plain (non-session) net calls under an f16x3 / f16 packing: raise when the call drove an activation beyond +-65504
    (its output then holds inf / NaN).  `seen`: the counter before the call; returns the counter now."""
    now = f16s_overflow_count()
    if now > seen:
        raise DfvoError("f16 split out of range: %d activation group(s) beyond +-65504 in %s -- its output holds inf / NaN; pack "
                        "the nets in exact fp32 (DFVO_CONV_PRECISION=fp32 or dfvo_hip.conv_precision: fp32)" % (now - seen, what))
    return now


def require_gpu():
    if lib().dfvo_device_count() < 1:
        raise DfvoError("no HIP device visible: the DF-VO hot path has no CPU fallback")


def set_sklearn_compat(version):
    """which scikit-learn the scale-recovery RANSAC reproduces where versions differ ("0.20.3": the reference's pin and the
    library default; pass sklearn.__version__ to match an installed one)"""
    check(lib().dfvo_set_sklearn_compat(str(version).encode()))


def f16s_overflow_count(reset=False):
    """events (threads / packed weights) that hit the +-65504 saturation of the f16x3 split since the last reset"""
    n = C.c_ulonglong(0)
    check(lib().dfvo_f16s_overflow_count(C.byref(n), int(bool(reset))))
    return int(n.value)


def as_ptr(a):
    """host pointer of a C-contiguous numpy array (or None)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def set_params(setter, handle, tensors):
    """push a {name: array-like} dict through dfvo_*_set_param"""
    for name, t in tensors.items():
        a = np.ascontiguousarray(np.asarray(t, dtype=np.float32))
        shape = (C.c_int * max(a.ndim, 1))(*a.shape)
        check(setter(handle, name.encode(), as_ptr(a), a.ndim, shape))
