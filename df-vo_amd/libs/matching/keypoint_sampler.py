"""KeypointSampler with the reference's surface (/root/reference/libs/matching/keypoint_sampler.py:18-163)
over dfvo_kp_local_bestn (local_bestN, kp_selection.py:74-200)."""
import ctypes as C

import numpy as np

from ... import capi
from ..tracker import _ctx


class KeypointSampler:
    def __init__(self, cfg):
        self.cfg = cfg
        self.kps = {}
        ks = self.cfg.kp_selection
        if ks.sampled_kp.enable or ks.bestN.enable:
            raise NotImplementedError("kp_selection.sampled_kp / bestN are ablation selectors "
                                      "(SURVEY.md section 8f rank 3); only local_bestN runs on the device")
        if ks.local_bestN.enable and ks.local_bestN.score_method != "flow":
            raise NotImplementedError("local_bestN.score_method '%s'" % ks.local_bestN.score_method)
        if ks.depth_consistency.enable:
            raise NotImplementedError("depth_consistency is experiment-only in the reference (out of scope)")

    def kp_selection(self, cur_data, ref_data):
        """keypoint_sampler.py:76-143 -> {'good_kp_found', 'kp1_best' [1,N,2], 'kp2_best' [1,N,2], 'fb_flow_mask'}"""
        outputs = {"good_kp_found": True}
        c = self.cfg.kp_selection.local_bestN
        if not c.enable:
            return outputs
        flow = np.ascontiguousarray(ref_data['flow'], dtype=np.float32)
        diff = np.ascontiguousarray(ref_data['flow_diff'], dtype=np.float32)
        h, w = cur_data['depth'].shape
        assert flow.shape == (2, h, w) and diff.shape[:2] == (h, w)
        nmax = int(c.num_bestN)
        kp1 = np.zeros((nmax, 2))
        kp2 = np.zeros((nmax, 2))
        n, good = C.c_int(), C.c_int()
        capi.check(capi.lib().dfvo_kp_local_bestn(_ctx.tracker(), capi.as_ptr(flow), capi.as_ptr(diff.reshape(h, w)), h,
                                                  w, int(c.num_row), int(c.num_col), nmax, float(c.thre),
                                                  capi.as_ptr(kp1), capi.as_ptr(kp2), C.byref(n), C.byref(good)))
        if not good.value:
            print("Cannot find enough good keypoints!")
            outputs['good_kp_found'] = False
            outputs['kp1_best'] = {}
            outputs['kp2_best'] = {}
            return outputs
        outputs['kp1_best'] = kp1[None, :n.value].copy()
        outputs['kp2_best'] = kp2[None, :n.value].copy()
        outputs['fb_flow_mask'] = diff.reshape(h, w)
        return outputs

    def update_kp_data(self, cur_data, ref_data, kp_sel_outputs):
        """keypoint_sampler.py:145-163"""
        if self.cfg.kp_selection.local_bestN.enable:
            ref_data['kp_best'] = kp_sel_outputs['kp1_best'][0]
            cur_data['kp_best'] = kp_sel_outputs['kp2_best'][0]
            cur_data['fb_flow_mask'] = kp_sel_outputs['fb_flow_mask']
