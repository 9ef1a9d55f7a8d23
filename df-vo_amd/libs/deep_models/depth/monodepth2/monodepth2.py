"""Monodepth2DepthNet with the reference's surface
(/root/reference/libs/deep_models/depth/monodepth2/monodepth2.py:22-139) over the dfvo_depthnet_* C ABI."""
import ctypes as C
import os

import numpy as np
import torch

from ..... import capi


class Monodepth2DepthNet:
    def __init__(self, height, width):
        self.height = height
        self.width = width
        self.device = torch.device('cuda')
        self.enable_finetune = False
        self.depth_scales = [0]
        self.model = None
        self.pred_depths = {}
        self.pred_disps = {}

    def initialize_network_model(self, weight_path, dataset, finetune):
        """monodepth2.py:30-89: directory with encoder.pth + depth.pth, or a dict
        {'encoder': state_dict (with 'height','width'), 'decoder': state_dict}"""
        if finetune:
            raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")
        if isinstance(weight_path, dict):
            enc, dec = weight_path['encoder'], weight_path['decoder']
        else:
            enc = torch.load(os.path.join(weight_path, 'encoder.pth'), map_location="cpu")
            dec = torch.load(os.path.join(weight_path, 'depth.pth'), map_location="cpu")
        self.feed_height = int(enc['height'])
        self.feed_width = int(enc['width'])
        if 'tum' in dataset:
            self.min_depth, self.max_depth, self.stereo_baseline_multiplier = 0.1, 10, 1
        else:
            self.min_depth, self.max_depth, self.stereo_baseline_multiplier = 0.1, 100, 5.4
        params = {}
        for k, v in enc.items():
            if hasattr(v, "detach") and v.dim() > 0 and "num_batches_tracked" not in k and not k.startswith("encoder.fc"):
                params[k] = v.detach().cpu().float().numpy()
        for k, v in dec.items():
            params[k] = v.detach().cpu().float().numpy()
        lib = capi.lib()
        capi.require_gpu()
        h = C.c_void_p()
        capi.check(lib.dfvo_depthnet_create(self.feed_height, self.feed_width, float(self.min_depth), float(self.max_depth),
                                            float(self.stereo_baseline_multiplier), None, C.byref(h)))
        capi.set_params(lib.dfvo_depthnet_set_param, h, params)
        capi.check(lib.dfvo_depthnet_finalize(h))
        self.model = h

    def inference_depth_u8(self, img_u8):
        """uint8 [feed_h, feed_w, 3] -> float32 [feed_h, feed_w] depth (x baseline multiplier)"""
        assert img_u8.shape == (self.feed_height, self.feed_width, 3) and img_u8.dtype == np.uint8
        depth = np.zeros((self.feed_height, self.feed_width), np.float32)
        capi.check(capi.lib().dfvo_depthnet_forward_host(self.model, capi.as_ptr(np.ascontiguousarray(img_u8)),
                                                         capi.as_ptr(depth)))
        return depth

    def inference_depth_image_u8(self, img_u8):
        """uint8 [H, W, 3] of any size -> float32 [feed_h, feed_w]: the Pillow-exact LANCZOS resize to the feed size
        (deep_models.py:195-199) runs on the device in front of the net"""
        assert img_u8.ndim == 3 and img_u8.shape[2] == 3 and img_u8.dtype == np.uint8
        depth = np.zeros((self.feed_height, self.feed_width), np.float32)
        capi.check(capi.lib().dfvo_depthnet_forward_image_host(self.model, capi.as_ptr(np.ascontiguousarray(img_u8)),
                                                               int(img_u8.shape[0]), int(img_u8.shape[1]),
                                                               capi.as_ptr(depth)))
        return depth

    def inference_depth(self, img):
        """monodepth2.py:124-139: [1,3,h,w] float tensor in [0,1] (ToTensor of a uint8 image) -> [1,1,h,w]"""
        a = img.detach().cpu().numpy()[0].transpose(1, 2, 0).astype(np.float64) * 255.0
        u = np.rint(a)
        if np.abs(a - u).max() > 1e-3:
            raise capi.DfvoError("Monodepth2DepthNet.inference_depth expects images quantised to k/255")
        d = self.inference_depth_u8(np.ascontiguousarray(u.astype(np.uint8)))
        out = torch.from_numpy(d)[None, None]
        self.pred_depths = {0: out / self.stereo_baseline_multiplier}
        self.pred_disps = {0: 1.0 / self.pred_depths[0]}
        return out

    def inference(self, img):
        """monodepth2.py:91-122: {'depth': {0: [1,1,h,w]}, 'disp': {0: [1,1,h,w]}} at the input size, depth WITHOUT the stereo
        baseline multiplier (monodepth2's 0.1-unit baseline), disp = the scaled disparity min_disp + (max_disp - min_disp) sigma.
        The device net returns the multiplied depth (what DF-VO consumes, inference_depth); both entries are derived from it
        on the host: depth = that / multiplier, disp = 1 / depth -- equal to the reference's tensors to float32 rounding, not bit
        for bit (the reference forms the disparity first and inverts it)."""
        self.inference_depth(img)
        return {'depth': dict(self.pred_depths), 'disp': dict(self.pred_disps)}

    def inference_no_grad(self, img):
        """deep_depth.py:74-85"""
        return self.inference(img)

    def setup_train(self, deep_model, cfg):
        raise NotImplementedError("online finetuning is out of scope of the inference hot path")
