import numpy as np
import torch


class ToTensor:
    """PIL image / uint8 HWC array -> float CHW in [0,1] (torchvision.transforms.ToTensor)"""

    def __call__(self, pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous()
        if t.dtype == torch.uint8:
            return t.float().div(255)
        return t
