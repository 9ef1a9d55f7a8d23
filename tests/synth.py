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


def two_view(n, out_frac=0.3, noise=0.15, seed=2002, w=1241, h=376):
    """seeded 3-D points seen from two poses: returns x1, x2 [n,2] pixels, R, t, K, outlier flags"""
    r = np.random.Generator(np.random.PCG64(seed))
    X = np.stack([r.uniform(-20, 20, n), r.uniform(-3, 3, n), r.uniform(5, 60, n)], 1)
    f = 718.856 * w / 1241.0
    K = np.array([[f, 0, 607.19 * w / 1241.0], [0, f, 185.22 * h / 376.0], [0, 0, 1]])
    wv = np.array([0.002, 0.01, 0.001])
    th = np.linalg.norm(wv)
    k = wv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.02, 0.01, 0.8])
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:]
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T
    x2 = x2[:, :2] / x2[:, 2:]
    x1 = x1 + r.normal(0, noise, x1.shape)
    x2 = x2 + r.normal(0, noise, x2.shape)
    o = r.random(n) < out_frac
    x2[o] = np.stack([r.uniform(0, w, int(o.sum())), r.uniform(0, h, int(o.sum()))], 1)
    return np.ascontiguousarray(x1), np.ascontiguousarray(x2), R, t, K, o
