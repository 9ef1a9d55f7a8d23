"""KeypointSampler with the reference's surface (/root/reference/libs/matching/keypoint_sampler.py:18-163)
over dfvo_kp_local_bestn (local_bestN, kp_selection.py:74-200), dfvo_kp_bestn (bestN_flow_kp, :33-71) and
dfvo_kp_sampled (sampled_kp, :327-378)."""
import ctypes as C

import numpy as np

from ... import capi
from ..tracker import _ctx


class KeypointSampler:
    def __init__(self, cfg):
        self.cfg = cfg
        self.kps = {}
        ks = self.cfg.kp_selection
        if ks.sampled_kp.enable:  # keypoint_sampler.py:29-36
            self.kps['uniform'] = self.generate_kp_samples(img_h=self.cfg.image.height, img_w=self.cfg.image.width,
                                                           crop=self.cfg.crop.flow_crop, N=ks.sampled_kp.num_kp)
        if ks.local_bestN.enable and ks.local_bestN.score_method not in ("flow", "flow_ratio"):
            raise NotImplementedError("local_bestN.score_method '%s' (flow_depth needs depth_consistency: Experiment Ver. only)"
                                      % ks.local_bestN.score_method)
        if ks.depth_consistency.enable:
            raise NotImplementedError("depth_consistency is experiment-only in the reference (out of scope)")
        # the frame session runs local_bestN behind the flow net with this configuration (libs/deep_models/session.py)
        self._session_kp_cfg = None
        if ks.local_bestN.enable:
            c = ks.local_bestN
            self._session_kp_cfg = capi.SessionKpCfg(num_row=int(c.num_row), num_col=int(c.num_col), num_bestN=int(c.num_bestN),
                                                     thre=float(c.thre), score_method={"flow": 0, "flow_ratio": 1}[c.score_method])
        _ctx.register_kp_cfg(self._session_kp_cfg)

    def get_feat_track_methods(self, method_idx):
        """keypoint_sampler.py:38-50: the one feature-tracking method of the release"""
        return {1: "deep_flow"}[method_idx]

    def generate_kp_samples(self, img_h, img_w, crop, N):
        """keypoint_sampler.py:51-74"""
        y0, y1 = int(crop[0][0] * img_h), int(crop[0][1] * img_h)
        x0, x1 = int(crop[1][0] * img_w), int(crop[1][1] * img_w)
        total_num = (x1 - x0) * (y1 - y0) - 1
        return np.linspace(0, total_num, N, dtype=int)

    def kp_selection(self, cur_data, ref_data):
        """keypoint_sampler.py:76-143 -> {'good_kp_found', 'kp1_best' [1,N,2], 'kp2_best' [1,N,2], 'fb_flow_mask'}
        and / or {'kp1_list', 'kp2_list'} [1,N,2] (sampled_kp)"""
        outputs = {"good_kp_found": True}
        if self.cfg.kp_selection.local_bestN.enable:
            outputs.update(self._local_bestN(cur_data, ref_data))
        elif self.cfg.kp_selection.bestN.enable:
            outputs.update(self._bestN(cur_data, ref_data))
        if self.cfg.kp_selection.sampled_kp.enable:
            outputs.update(self._sampled_kp(cur_data, ref_data))
        return outputs

    def _bestN(self, cur_data, ref_data):
        """bestN_flow_kp (kp_selection.py:33-71) on the device (dfvo_kp_bestn)"""
        flow = np.ascontiguousarray(ref_data['flow'], dtype=np.float32)
        diff = np.ascontiguousarray(ref_data['flow_diff'], dtype=np.float32)
        h, w = cur_data['depth'].shape
        assert flow.shape == (2, h, w) and diff.shape[:2] == (h, w)
        N = int(self.cfg.kp_selection.bestN.num_bestN)
        kp1, kp2 = np.zeros((N, 2)), np.zeros((N, 2))
        n = C.c_int()
        capi.check(capi.lib().dfvo_kp_bestn(_ctx.tracker_exclusive(), capi.as_ptr(flow), capi.as_ptr(diff.reshape(h, w)), h, w, N,
                                            capi.as_ptr(kp1), capi.as_ptr(kp2), C.byref(n)))
        if n.value != N:
            raise ValueError("kth(=%d) out of bounds (%d)" % (N, h * w))  # what np.argpartition raises in the reference
        return {"kp1_best": kp1[None], "kp2_best": kp2[None], "fb_flow_mask": diff.reshape(h, w)}

    def _sampled_kp(self, cur_data, ref_data):
        """kp_selection.py:327-378 on the device (dfvo_kp_sampled)"""
        flow = np.ascontiguousarray(ref_data['flow'], dtype=np.float32)
        h, w = ref_data['depth'].shape
        assert flow.shape == (2, h, w)
        crop = self.cfg.crop.flow_crop
        y0, y1, x0, x1 = 0, h, 0, w
        if crop is not None:
            y0, y1 = int(h * crop[0][0]), int(h * crop[0][1])
            x0, x1 = int(w * crop[1][0]), int(w * crop[1][1])
        idx = np.ascontiguousarray(self.kps['uniform'], dtype=np.int32)
        kp1 = np.zeros((len(idx), 2))
        kp2 = np.zeros((len(idx), 2))
        capi.check(capi.lib().dfvo_kp_sampled(_ctx.tracker_exclusive(), capi.as_ptr(flow), h, w, y0, y1, x0, x1, capi.as_ptr(idx),
                                              len(idx), capi.as_ptr(kp1), capi.as_ptr(kp2)))
        return {"kp1_list": kp1[None], "kp2_list": kp2[None]}

    def _local_bestN(self, cur_data, ref_data):
        outputs = {"good_kp_found": True}
        c = self.cfg.kp_selection.local_bestN
        s = _ctx.session
        if s is not None and self._session_kp_cfg is not None:
            # the arrays are (copies of) what forward_flow returned for the pair the session holds: the selection already ran
            # on the device-resident flow, behind the flow net
            res = s.keypoints(ref_data['flow'], ref_data['flow_diff'], self._session_kp_cfg)
            s.stats["kp_resident" if res is not None else "kp_plain"] += 1
            if res is not None:
                kp1, kp2, n, good = res
                h, w = cur_data['depth'].shape
                assert ref_data['flow'].shape == (2, h, w)
                if not good:
                    print("Cannot find enough good keypoints!")
                    return {"good_kp_found": False, "kp1_best": {}, "kp2_best": {}}
                return {"good_kp_found": True, "kp1_best": np.array(kp1)[None], "kp2_best": np.array(kp2)[None],
                        "fb_flow_mask": np.asarray(ref_data['flow_diff'], dtype=np.float32).reshape(h, w)}
        flow = np.ascontiguousarray(ref_data['flow'], dtype=np.float32)
        diff = np.ascontiguousarray(ref_data['flow_diff'], dtype=np.float32)
        h, w = cur_data['depth'].shape
        assert flow.shape == (2, h, w) and diff.shape[:2] == (h, w)
        nmax = int(c.num_bestN)
        kp1 = np.zeros((nmax, 2))
        kp2 = np.zeros((nmax, 2))
        n, good = C.c_int(), C.c_int()
        score = {"flow": 0, "flow_ratio": 1}[c.score_method]
        capi.check(capi.lib().dfvo_kp_local_bestn_ex(_ctx.tracker_exclusive(), capi.as_ptr(flow), capi.as_ptr(diff.reshape(h, w)), h,
                                                     w, int(c.num_row), int(c.num_col), nmax, float(c.thre), score,
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
        if self.cfg.kp_selection.local_bestN.enable or self.cfg.kp_selection.bestN.enable:
            ref_data['kp_best'] = kp_sel_outputs['kp1_best'][0]
            cur_data['kp_best'] = kp_sel_outputs['kp2_best'][0]
            cur_data['fb_flow_mask'] = kp_sel_outputs['fb_flow_mask']
        if self.cfg.kp_selection.sampled_kp.enable:
            ref_data['kp_list'] = kp_sel_outputs['kp1_list'][0]
            cur_data['kp_list'] = kp_sel_outputs['kp2_list'][0]
