"""LiteFlow with the reference's surface (/root/reference/libs/deep_models/flow/lite_flow_net/lite_flow.py:23-148
and the DeepFlow base, flow/deep_flow.py:22-196) over the dfvo_flownet_* C ABI: the whole LiteFlowNet
forward/backward pass, the resizes and the consistency map run in HIP kernels."""
import ctypes as C
import math

import numpy as np
import torch

from ..... import capi


class LiteFlow:
    def __init__(self, height, width):
        self.height = height
        self.width = width
        self.batch_size = 1
        self.device = torch.device('cuda')
        self.enable_finetune = False
        self.flow_scales = [1]
        self.half_flow = False
        self.model = None          # opaque dfvo_flownet handle (the reference holds a torch module here)
        self.session = None        # FrameSession sharing this net (DeepModel.initialize_models)
        self.forward_flow = {}
        self.backward_flow = {}
        self.flow_diff = {}
        self.px1on2 = {}

    # -- DeepFlow helpers kept for API parity -------------------------------------------------
    def get_target_size(self, h, w):
        """deep_flow.py:89-105, evaluated by the library exactly as the reference evaluates it (dfvo_flow_target_size)"""
        import ctypes as C
        th, tw = C.c_int(), C.c_int()
        capi.check(capi.lib().dfvo_flow_target_size(int(h), int(w), C.byref(th), C.byref(tw)))
        return th.value, tw.value

    def resize_dense_flow(self, flow, des_height, des_width):
        """deep_flow.py:107-129: [N,2,H,W] flow tensor -> [N,2,H',W'], bilinear with aligned corners (dfvo_resize_bilinear on the
        device), x / y components scaled by the width / height ratio"""
        n, c, h, w = flow.shape
        assert c == 2
        capi.require_gpu()
        x = torch.zeros((n, h, w, 4), dtype=torch.float32, device="cuda")
        x[..., :2] = flow.detach().float().permute(0, 2, 3, 1)
        out = torch.empty((n, int(des_height), int(des_width), 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        capi.check(capi.lib().dfvo_resize_bilinear(C.c_void_p(x.data_ptr()), n, h, w, 4, C.c_void_p(out.data_ptr()),
                                                   int(des_height), int(des_width), 1, None))
        capi.check(capi.lib().dfvo_sync_device())
        res = out[..., :2].permute(0, 3, 1, 2).cpu()
        return torch.stack([res[:, 0] * float(des_width / w), res[:, 1] * float(des_height / h)], dim=1)

    def forward_backward_consistency(self, flow1, flow2, px1on2):
        """deep_flow.py:171-196: |flow1 - grid_sample(-flow2, px1on2)| -> [N,H,W,1].  Inside inference_flow the library
        computes this map itself (k_flow_consistency); this stand-alone form samples through dfvo_backward_warp: the
        normalised grid is handed over as a displacement from a zero base grid, so the sample positions agree with torch's
        to one rounding of the grid coordinate, not bit for bit."""
        n, c, h, w = flow2.shape
        assert c == 2 and tuple(px1on2.shape) == (n, h, w, 2)
        capi.require_gpu()
        src = torch.zeros((n, h, w, 4), dtype=torch.float32, device="cuda")
        src[..., :2] = (-flow2.detach().float()).permute(0, 2, 3, 1)
        disp = px1on2.detach().float().clone()
        disp[..., 0] *= (w - 1.0) / 2.0
        disp[..., 1] *= (h - 1.0) / 2.0
        disp = disp.contiguous().cuda()
        dst = torch.empty_like(src)
        zx, zy = np.zeros(w, np.float32), np.zeros(h, np.float32)
        torch.cuda.synchronize()
        capi.check(capi.lib().dfvo_backward_warp(C.c_void_p(src.data_ptr()), C.c_void_p(disp.data_ptr()), 1.0, n, h, w, 4,
                                                 capi.as_ptr(zx), capi.as_ptr(zy), C.c_void_p(dst.data_ptr()), None))
        capi.check(capi.lib().dfvo_sync_device())
        warp = dst[..., :2].permute(0, 3, 1, 2).cpu()
        return (flow1.detach().float().cpu() - warp).norm(dim=1, keepdim=True).permute(0, 2, 3, 1)

    def load_flow_file(self, flow_path):
        raise NotImplementedError("pre-computed flow files are a dataset-loader path of the reference (deep_flow.py:131-155); "
                                  "out of scope of the tracking hot path")

    def initialize_network_model(self, weight_path, finetune):
        """lite_flow.py:32-53: `weight_path` is a torch state_dict file, or an in-memory state_dict"""
        if finetune:
            raise NotImplementedError("online finetuning is training; out of scope of the inference hot path")
        if weight_path is None:
            assert False, "No LiteFlowNet pretrained model is provided."
        sd = weight_path if isinstance(weight_path, dict) else torch.load(weight_path, map_location="cpu")
        lib = capi.lib()
        capi.require_gpu()
        h = C.c_void_p()
        capi.check(lib.dfvo_flownet_create(int(self.height), int(self.width), None, C.byref(h)))
        nh, nw = C.c_int(), C.c_int()
        capi.check(lib.dfvo_flownet_net_size(h, C.byref(nh), C.byref(nw)))
        params = {k: (v.detach().cpu().float().numpy() if hasattr(v, "detach") else np.asarray(v, np.float32))
                  for k, v in sd.items()}
        for l in range(1, 7):  # torch's own linspace tables keep Backward() bit-compatible with the oracle
            params["aux.linspace_x.%d" % l] = torch.linspace(-1.0, 1.0, nw.value >> (l - 1)).numpy()
            params["aux.linspace_y.%d" % l] = torch.linspace(-1.0, 1.0, nh.value >> (l - 1)).numpy()
        capi.set_params(lib.dfvo_flownet_set_param, h, params)
        capi.check(lib.dfvo_flownet_finalize(h))
        self.model = h
        self.net_size = (nh.value, nw.value)

    # -- inference ---------------------------------------------------------------------------------
    def _to_u8(self, img):
        """[1,3,H,W] float tensor in [0,1] (deep_models.py:160-163) -> uint8 HWC; must be exact k/255 values"""
        a = img.detach().cpu().numpy()[0].transpose(1, 2, 0).astype(np.float64) * 255.0
        u = np.rint(a)
        if np.abs(a - u).max() > 1e-3:
            raise capi.DfvoError("LiteFlow.inference_flow expects images quantised to k/255 (uint8 frames)")
        return np.ascontiguousarray(u.astype(np.uint8))

    def inference_flow_u8(self, ref_u8, cur_u8):
        """uint8 HWC frames -> numpy fwd [2,H,W], bwd [2,H,W], diff [H,W,1] (float32)"""
        h, w = self.height, self.width
        assert ref_u8.shape == (h, w, 3) and cur_u8.shape == (h, w, 3) and ref_u8.dtype == np.uint8
        fwd = np.zeros((2, h, w), np.float32)
        bwd = np.zeros((2, h, w), np.float32)
        diff = np.zeros((h, w, 1), np.float32)
        if self.session is not None:  # this pass overwrites the pyramids the session would carry over
            self.session.invalidate_carry()
        capi.check(capi.lib().dfvo_flownet_forward_host(self.model, capi.as_ptr(np.ascontiguousarray(ref_u8)),
                                                        capi.as_ptr(np.ascontiguousarray(cur_u8)), capi.as_ptr(fwd),
                                                        capi.as_ptr(bwd), capi.as_ptr(diff)))
        return fwd, bwd, diff

    def inference_flow(self, img1, img2, forward_backward=False, dataset='kitti'):
        """lite_flow.py:89-148 -> {'forward' [1,2,H,W], 'backward', 'flow_diff' [1,H,W,1]} torch tensors"""
        fwd, bwd, diff = self.inference_flow_u8(self._to_u8(img1), self._to_u8(img2))
        flows = {'forward': torch.from_numpy(fwd)[None]}
        self.forward_flow = {1: flows['forward']}
        if forward_backward:
            flows['backward'] = torch.from_numpy(bwd)[None]
            flows['flow_diff'] = torch.from_numpy(diff)[None]
            self.backward_flow = {1: flows['backward']}
            self.flow_diff = {1: flows['flow_diff']}
        return flows

    def inference(self, img1, img2):
        return {1: self.inference_flow(img1, img2, forward_backward=False)['forward']}

    inference_no_grad = inference

    def setup_train(self, deep_model, cfg):
        raise NotImplementedError("online finetuning is out of scope of the inference hot path")
