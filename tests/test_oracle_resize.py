"""CPU: the oracle's restatement of cv2.resize for the loaded uint8 frame (utils.py:51, default INTER_LINEAR; OpenCV
3.4.3's 11-bit fixed point).  OpenCV is not installed here, so the restatement is checked through what it must satisfy
(tools/opencv343_dump.py writes the real library's outputs for the same cases where opencv-python 3.4.3 exists)."""
import numpy as np
import pytest

from oracle import cv2_shim as cv2


def resize_case(seed, h, w, c=3):
    r = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 80 * np.sin(xx[..., None] / 17.0 + np.arange(c)) * np.cos(yy[..., None] / 11.0)
    img = np.clip(base + r.normal(0, 25, (h, w, c)), 0, 255).astype(np.uint8)
    return img if c > 1 else img[..., 0]


def float_bilinear(img, ow, oh):
    """half-pixel-centre bilinear interpolation in float64 with edge clamping"""
    h, w = img.shape[:2]
    im = img.reshape(h, w, -1).astype(np.float64)
    fx = (np.arange(ow) + 0.5) * (w / ow) - 0.5
    fy = (np.arange(oh) + 0.5) * (h / oh) - 0.5
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    ax, ay = fx - x0, fy - y0
    xa, xb = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    ya, yb = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    top = im[ya][:, xa] * (1 - ax)[None, :, None] + im[ya][:, xb] * ax[None, :, None]
    bot = im[yb][:, xa] * (1 - ax)[None, :, None] + im[yb][:, xb] * ax[None, :, None]
    return top * (1 - ay)[:, None, None] + bot * ay[:, None, None]


SIZES = [(376, 1241, 192, 640), (370, 1226, 192, 640), (192, 640, 376, 1241), (480, 640, 256, 320), (37, 53, 90, 100),
         (960, 1280, 256, 640), (5, 7, 3, 2)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_linear_resize_stays_within_one_grey_level_of_float_bilinear(h, w, oh, ow):
    img = resize_case(h * 7 + w, h, w)
    out = cv2.resize(img, (ow, oh))
    assert out.shape == (oh, ow, 3) and out.dtype == np.uint8
    ref = float_bilinear(img, ow, oh)
    err = np.abs(out.astype(np.float64) - ref)
    # 11-bit weights, two truncating shifts (each drops up to a quarter grey level) and a rounding add: below one grey
    # level everywhere, and the small negative bias this arithmetic is known for
    assert err.max() < 1.0, err.max()
    bias = (out.astype(np.float64) - ref).mean()
    assert -0.3 < bias <= (0.01 if oh * ow > 1000 else 0.2), bias


def test_linear_resize_fixed_points():
    img = resize_case(5, 60, 80)
    assert np.array_equal(cv2.resize(img, (80, 60)), img)                       # same size: every weight is (2048, 0)
    assert np.unique(cv2.resize(np.full((37, 53, 3), 77, np.uint8), (100, 90))).tolist() == [77]
    assert np.unique(cv2.resize(np.full((37, 53), 255, np.uint8), (20, 11))).tolist() == [255]
    g = np.tile(np.arange(200, dtype=np.uint8)[None, :], (4, 1))
    assert cv2.resize(g, (400, 4))[0, :8].tolist() == [0, 0, 1, 1, 2, 2, 3, 3]  # 0, .25, .75, 1.25 ... rounded half up
    assert cv2.resize(g, (400, 4))[0, -3:].tolist() == [198, 199, 199]


def test_exact_half_size_is_the_2x2_mean():
    img = resize_case(9, 96, 128).astype(np.int64)
    want = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(cv2.resize(img.astype(np.uint8), (64, 48)), want.astype(np.uint8))
    # only BOTH axes at exactly 2 take that path
    out = cv2.resize(img.astype(np.uint8), (64, 50))
    assert out.shape == (50, 64, 3)


def test_nearest_still_served():
    d = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(cv2.resize(d, (8, 6), interpolation=cv2.INTER_NEAREST), d.repeat(2, 0).repeat(2, 1))


def test_known_answer_two_pixels_to_four():
    """cv2.resize(np.array([[0, 255]], np.uint8), (4, 1)) is [[0, 64, 191, 255]] in every OpenCV 3.x / 4.x (the textbook
    example of its half-pixel-centre convention and of the 11-bit fixed point: 63.75 and 191.25 round to 64 and 191)"""
    assert cv2.resize(np.array([[0, 255]], np.uint8), (4, 1)).tolist() == [[0, 64, 191, 255]]
    assert cv2.resize(np.array([[0], [255]], np.uint8), (1, 4)).ravel().tolist() == [0, 64, 191, 255]
