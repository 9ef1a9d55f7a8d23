"""GPU: dfvo_resize_linear_u8 (the device replacement of read_image's cv2.resize, utils.py:51) against the oracle's
restatement of OpenCV 3.4.3's 8-bit INTER_LINEAR arithmetic -- bit-exact, every size class the datasets produce."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from oracle import cv2_shim as cv2
from test_oracle_resize import SIZES, resize_case

pytestmark = pytest.mark.gpu


def dev_resize(gpu, img, oh, ow):
    c = 1 if img.ndim == 2 else img.shape[2]
    src = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    dst = torch.zeros((oh, ow, c), dtype=torch.uint8, device="cuda")
    gpu.check(gpu.lib().dfvo_resize_linear_u8(C.c_void_p(src.data_ptr()), img.shape[0], img.shape[1], c,
                                              C.c_void_p(dst.data_ptr()), oh, ow, None))
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    return out[..., 0] if img.ndim == 2 else out


@pytest.mark.parametrize("h,w,oh,ow", SIZES + [(96, 128, 48, 64), (96, 128, 50, 64), (376, 1241, 376, 1241), (1, 1, 4, 5),
                                                 (1280, 1920, 376, 1241), (2, 3, 1, 1)])
def test_linear_resize_bit_exact(gpu, h, w, oh, ow):
    img = resize_case(h * 7 + w, h, w)
    got = dev_resize(gpu, img, oh, ow)
    want = cv2.resize(img, (ow, oh))
    assert got.shape == want.shape
    assert np.array_equal(got, want), np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.parametrize("c", [1, 4])
def test_linear_resize_other_channel_counts(gpu, c):
    img = resize_case(77, 123, 217, c)
    assert np.array_equal(dev_resize(gpu, img, 64, 100), cv2.resize(img, (100, 64)))
    assert np.array_equal(dev_resize(gpu, img, 300, 400), cv2.resize(img, (400, 300)))


def test_image_to_device_is_read_image_after_decoding(gpu):
    """sequence.image_to_device = crop + resize of read_image (utils.py:46-51) with the frame already decoded"""
    smod = importlib.import_module("df-vo_amd.sequence")
    img = resize_case(3, 370, 1226)
    crop = [[0.2, 1.0], [0.05, 0.95]]
    got = smod.image_to_device(img, 192, 640, crop).cpu().numpy()
    y0, y1, x0, x1 = int(370 * 0.2), int(370 * 1.0), int(1226 * 0.05), int(1226 * 0.95)
    assert np.array_equal(got, cv2.resize(np.ascontiguousarray(img[y0:y1, x0:x1]), (640, 192)))
    assert np.array_equal(smod.image_to_device(img, 192, 640).cpu().numpy(), cv2.resize(img, (640, 192)))
    # the frame as cv2.imread leaves it (BGR): cvtColor + crop + resize in the one launch, crop read with the frame's pitch
    bgr = np.ascontiguousarray(img[..., ::-1])
    got = smod.image_to_device(bgr, 192, 640, crop, bgr=True).cpu().numpy()
    assert np.array_equal(got, cv2.resize(np.ascontiguousarray(img[y0:y1, x0:x1]), (640, 192)))
    assert np.array_equal(smod.image_to_device(bgr, 370, 1226, bgr=True).cpu().numpy(), img)     # same size: cv2.resize is the identity
    half = resize_case(3, 376, 1240)
    assert np.array_equal(smod.image_to_device(np.ascontiguousarray(half[..., ::-1]), 188, 620, bgr=True).cpu().numpy(),
                          cv2.resize(half, (620, 188)))                                                   # INTER_AREA's 2 x 2 fast path
