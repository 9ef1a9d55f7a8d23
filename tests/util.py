"""helpers shared by the tests"""
import numpy as np
import torch


def nhwc_dev(t, cs=None):
    """NCHW float CPU tensor -> NHWC cuda tensor with `cs` floats per pixel (zero padded)"""
    n, c, h, w = t.shape
    cs = cs or ((c + 3) // 4 * 4)
    out = torch.zeros(n, h, w, cs, dtype=torch.float32)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out.cuda().contiguous()


def nchw_host(d, c):
    """NHWC cuda tensor -> NCHW CPU tensor with the first c channels"""
    return d[..., :c].permute(0, 3, 1, 2).contiguous().cpu()


def ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def report(name, got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    scale = max(1e-30, float(np.abs(want).max()))
    print("%-40s max|err| %.3e  (max|ref| %.3e, rel %.3e)" % (name, err.max(), scale, err.max() / scale))
    return float(err.max()), scale
