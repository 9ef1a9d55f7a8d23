"""DeepModel with the reference's surface (/root/reference/libs/deep_models/deep_models.py:25-206):
builds the two nets and provides forward_depth / forward_flow with the reference's argument and
return conventions; everything, the PIL-LANCZOS resize of the depth input included, runs in libdfvo_hip.so."""
import os

import numpy as np
import torch

from ... import capi
from ..tracker import _ctx
from .depth.monodepth2.monodepth2 import Monodepth2DepthNet
from .flow.lite_flow_net.lite_flow import LiteFlow
from .session import FrameSession


def hip_options(cfg):
    """Options of this implementation that have no counterpart in the reference's YAML, read from an OPTIONAL top-level key
    `dfvo_hip` of the configuration (absent in every reference config file: the defaults below apply), environment first:
      conv_precision  "f16x3" (default here: fp32-class split products, the arithmetic bench.py's headline runs), "fp32"
                      (exact fp32 MFMA, the C library's own default) or "f16" (one product per term: not fp32-class);
                      DFVO_CONV_PRECISION overrides
      session         True (default): the frame session of libs/deep_models/session.py; DFVO_SESSION=0 turns it off"""
    o = {"conv_precision": "f16x3", "session": True}
    try:
        extra = cfg.get("dfvo_hip", None) if hasattr(cfg, "get") else getattr(cfg, "dfvo_hip", None)
    except Exception:  # (config objects without the key)
        extra = None
    if extra:
        o.update({k: extra[k] for k in o if k in extra})
    if os.environ.get("DFVO_CONV_PRECISION"):
        o["conv_precision"] = os.environ["DFVO_CONV_PRECISION"]
    if os.environ.get("DFVO_SESSION") is not None:
        o["session"] = os.environ["DFVO_SESSION"] != "0"
    return o


class DeepModel:
    def __init__(self, cfg):
        self.cfg = cfg
        self.finetune_cfg = self.cfg.online_finetune
        self.device = torch.device('cuda')

    def initialize_models(self):
        """deep_models.py:38-57.  The nets are packed in the precision hip_options() names -- f16x3 unless the configuration
        or the environment says otherwise: an unmodified apis/run.py gets the arithmetic the headline is measured in, not the
        C library's exact-fp32 default -- and, when both nets exist, the frame session is created over them."""
        opts = hip_options(self.cfg)
        self.conv_precision = opts["conv_precision"]
        lib = capi.lib()
        before = lib.dfvo_get_conv_precision()  # the process-wide setting is restored once these nets are packed
        capi.check(lib.dfvo_set_conv_precision(self.conv_precision.encode()))
        self.session = None
        seen = capi.f16s_overflow_count() if self.conv_precision != "fp32" else 0
        try:
            self.flow = self.initialize_deep_flow_model()
            if self.cfg.depth.depth_src is None:
                if self.cfg.depth.deep_depth.pretrained_model is not None:
                    self.depth = self.initialize_deep_depth_model()
                else:
                    assert False, "No precomputed depths nor pretrained depth model"
        finally:
            capi.check(lib.dfvo_set_conv_precision(before))
        if self.conv_precision != "fp32":
            # weights beyond f16's range are clamped (and counted) by the packer: refuse them here, loudly
            self._f16_seen = capi.check_f16_range(seen, "the packed weights")
        if self.cfg.deep_pose.enable:
            raise NotImplementedError("deep_pose is 'Experiment Ver. only' in the reference; out of scope")
        if opts["session"] and getattr(self, "depth", None) is not None and isinstance(self.flow, LiteFlow):
            self.session = FrameSession(self.flow.model, self.depth.model, _ctx.tracker(), self.cfg.image.height,
                                        self.cfg.image.width, self.depth.feed_height, self.depth.feed_width)
            self.flow.session = self.session
        _ctx.register_session(self.session)

    def initialize_deep_flow_model(self):
        """deep_models.py:59-84"""
        if self.cfg.deep_flow.network == 'liteflow':
            net = LiteFlow(self.cfg.image.height, self.cfg.image.width)
            net.initialize_network_model(weight_path=self.cfg.deep_flow.flow_net_weight,
                                         finetune=self.finetune_cfg.enable and self.finetune_cfg.flow.enable)
            return net
        if self.cfg.deep_flow.network == 'hd3':
            raise NotImplementedError("HD3 is an alternative flow net of the reference; out of scope")
        assert False, "Invalid flow network [{}] is provided.".format(self.cfg.deep_flow.network)

    def initialize_deep_depth_model(self):
        """deep_models.py:86-102"""
        if self.cfg.depth.deep_depth.network == 'monodepth2':
            net = Monodepth2DepthNet(self.cfg.image.height, self.cfg.image.width)
            net.initialize_network_model(weight_path=self.cfg.depth.deep_depth.pretrained_model,
                                         dataset=self.cfg.dataset,
                                         finetune=self.finetune_cfg.enable and self.finetune_cfg.depth.enable)
            return net
        assert False, "Invalid depth network [{}] is provided.".format(self.cfg.depth.deep_depth.network)

    def setup_train(self):
        raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")

    def forward_flow(self, in_cur_data, in_ref_data, forward_backward):
        """deep_models.py:144-182: flows[(ref,cur)], flows[(cur,ref)], flows[(ref,cur,'diff')]"""
        s = self.session
        if s is not None and s.holds_pair(in_ref_data['img'], in_cur_data['img']):
            fwd, bwd, diff = s.flow()  # enqueued when forward_depth pushed the current frame: wait, pinned views
            s.stats["flow_resident"] += 1
        else:
            if s is not None:
                s.stats["flow_plain"] += 1
            fwd, bwd, diff = self.flow.inference_flow_u8(np.ascontiguousarray(in_ref_data['img']),
                                                         np.ascontiguousarray(in_cur_data['img']))
            self._check_range("the flow net")
        src_id, tgt_id = in_ref_data['id'], in_cur_data['id']
        flows = {(src_id, tgt_id): fwd}
        if forward_backward:
            flows[(tgt_id, src_id)] = bwd
            flows[(src_id, tgt_id, "diff")] = diff
        return flows

    def forward_depth(self, imgs):
        """deep_models.py:184-206: LANCZOS resize to the feed size (Pillow's 8-bit arithmetic, on the device), then
        the net; only the raw frame crosses PCIe.  With the frame session this call is where the frame enters the device: the
        flow net of (previous frame, this frame) and the keypoint / homography stage behind it are enqueued here as well."""
        if self.session is not None and self.session.accepts(imgs[0]):
            return self.session.push(imgs[0])
        depth = self.depth.inference_depth_image_u8(np.ascontiguousarray(imgs[0]))
        self._check_range("the depth net")
        return depth

    def _check_range(self, what):
        """plain entry points under an f16x3 / f16 packing: fail instead of handing out inf / NaN (the session does the same
        from the counter it reads behind each net, libs/deep_models/session.py)"""
        if self.conv_precision != "fp32":
            self._f16_seen = capi.check_f16_range(getattr(self, "_f16_seen", 0), what)

    def initialize_deep_pose_model(self):
        raise NotImplementedError("deep_pose is 'Experiment Ver. only' in the reference; out of scope")

    def forward_pose(self, imgs):
        raise NotImplementedError("deep_pose is 'Experiment Ver. only' in the reference; out of scope")

    def finetune(self, img1, img2, pose, K, inv_K):
        raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")

    def save_model(self):
        raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")
