"""seeded synthetic inputs"""
import numpy as np


def smooth_noise(rng, h, w, sigmas=(2, 4, 8, 16)):
    from scipy.ndimage import gaussian_filter
    acc = np.zeros((h, w, 3))
    for s in sigmas:
        n = rng.standard_normal((h, w, 3))
        f = gaussian_filter(n, sigma=(s, s, 0), mode="reflect")
        acc += f / f.std()
    return acc


def image_pair(h, w, seed=1001, shift=(1.5, -0.8)):
    """textured uint8 RGB frame + a second frame = the first warped by a smooth flow field"""
    from scipy.ndimage import map_coordinates
    rng = np.random.Generator(np.random.PCG64(seed))
    base = smooth_noise(rng, h + 32, w + 32)
    img1 = np.clip(128 + 48 * base[16:16 + h, 16:16 + w] / 2.0, 0, 255)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    # flow grows towards the image bottom (planar-road like)
    fx = shift[0] * (1 + 2.0 * yy / h)
    fy = shift[1] * (1 + 1.0 * yy / h)
    img2 = np.stack([map_coordinates(128 + 48 * base[..., c] / 2.0, [yy + 16 - fy, xx + 16 - fx], order=1, mode="nearest")
                     for c in range(3)], -1)
    return img1.astype(np.uint8), np.clip(img2, 0, 255).astype(np.uint8)
