"""one shared dfvo_tracker handle (device workspaces + the device copy of the numpy RandomState)"""
import ctypes as C

import numpy as np

from ... import capi

_trk = None
session = None       # the FrameSession of the process's DeepModel (libs/deep_models/session.py); None: plain entry points
_kp_cfg = None       # capi.SessionKpCfg of the KeypointSampler, registered at its construction
_pose_cfg_fn = None  # () -> capi.Pose2d2dCfg of the EssTracker


def register_session(s):
    """DeepModel.initialize_models: the session picks up whatever the sampler / tracker registered before or after it"""
    global session
    if session is not None and session is not s:
        session.retire()  # one speculating session per tracker: the older one's DeepModel continues on the plain entry points
    session = s
    if s is not None:
        s.kp_cfg = _kp_cfg
        s.pose_cfg_fn = _pose_cfg_fn
        s.rng_words = numpy_rng_words


def register_kp_cfg(c):
    global _kp_cfg
    _kp_cfg = c
    if session is not None:
        session.kp_cfg = c


def register_pose_cfg(fn):
    global _pose_cfg_fn
    _pose_cfg_fn = fn
    if session is not None:
        session.pose_cfg_fn = fn


def tracker():
    global _trk
    if _trk is None:
        capi.require_gpu()
        h = C.c_void_p()
        capi.check(capi.lib().dfvo_tracker_create(None, C.byref(h)))
        _trk = h
    return _trk


def tracker_exclusive():
    """the tracker handle for a plain (host-array) solver entry point: the frame session may still have its speculative
    keypoint / homography stage in flight on the tracker's buffers -- it is waited for and abandoned first"""
    if session is not None:
        session.quiesce()
    return tracker()


def numpy_rng_words():
    """np.random's global RandomState (the stream the reference consumes) as the 625 words the C ABI takes: key | position"""
    st = np.random.get_state()
    s = np.empty(625, np.uint32)
    s[:624] = st[1]
    s[624] = st[2]
    return s


def push_numpy_rng():
    """upload np.random's global RandomState to the device"""
    s = numpy_rng_words()
    capi.check(capi.lib().dfvo_tracker_set_rng_state(tracker(), capi.as_ptr(s)))


def set_numpy_rng(s):
    """hand an advanced stream (625 words) back to numpy (has_gauss / cached_gaussian are untouched by integer draws)"""
    st = np.random.get_state()
    np.random.set_state((st[0], s[:624].copy(), int(s[624]), st[3], st[4]))


def pull_numpy_rng():
    """read the tracker's advanced stream back and hand it to numpy"""
    s = np.zeros(625, np.uint32)
    capi.check(capi.lib().dfvo_tracker_get_rng_state(tracker(), capi.as_ptr(s)))
    set_numpy_rng(s)
