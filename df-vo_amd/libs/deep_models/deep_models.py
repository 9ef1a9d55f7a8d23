"""DeepModel with the reference's surface (/root/reference/libs/deep_models/deep_models.py:25-206):
builds the two nets and provides forward_depth / forward_flow with the reference's argument and
return conventions; everything, the PIL-LANCZOS resize of the depth input included, runs in libdfvo_hip.so."""
import numpy as np
import torch

from .depth.monodepth2.monodepth2 import Monodepth2DepthNet
from .flow.lite_flow_net.lite_flow import LiteFlow


class DeepModel:
    def __init__(self, cfg):
        self.cfg = cfg
        self.finetune_cfg = self.cfg.online_finetune
        self.device = torch.device('cuda')

    def initialize_models(self):
        """deep_models.py:38-57"""
        self.flow = self.initialize_deep_flow_model()
        if self.cfg.depth.depth_src is None:
            if self.cfg.depth.deep_depth.pretrained_model is not None:
                self.depth = self.initialize_deep_depth_model()
            else:
                assert False, "No precomputed depths nor pretrained depth model"
        if self.cfg.deep_pose.enable:
            raise NotImplementedError("deep_pose is 'Experiment Ver. only' in the reference; out of scope")

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
        fwd, bwd, diff = self.flow.inference_flow_u8(np.ascontiguousarray(in_ref_data['img']),
                                                     np.ascontiguousarray(in_cur_data['img']))
        src_id, tgt_id = in_ref_data['id'], in_cur_data['id']
        flows = {(src_id, tgt_id): fwd}
        if forward_backward:
            flows[(tgt_id, src_id)] = bwd
            flows[(src_id, tgt_id, "diff")] = diff
        return flows

    def forward_depth(self, imgs):
        """deep_models.py:184-206: LANCZOS resize to the feed size (Pillow's 8-bit arithmetic, on the device), then
        the net; only the raw frame crosses PCIe"""
        return self.depth.inference_depth_image_u8(np.ascontiguousarray(imgs[0]))

    def initialize_deep_pose_model(self):
        raise NotImplementedError("deep_pose is 'Experiment Ver. only' in the reference; out of scope")

    def forward_pose(self, imgs):
        raise NotImplementedError("deep_pose is 'Experiment Ver. only' in the reference; out of scope")

    def finetune(self, img1, img2, pose, K, inv_K):
        raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")

    def save_model(self):
        raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")
