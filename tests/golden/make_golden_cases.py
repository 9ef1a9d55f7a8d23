"""Seeded case generators of the fixtures that code WITHOUT torch must regenerate too (tests/second_env_ref_tracker.py runs
under an interpreter that has none); make_golden.py re-exports them."""
import numpy as np


PNP_CASES = {"a": (51, 2000, 0.3, 0.2, True, 0), "b": (52, 600, 0.6, 0.5, False, 0), "c": (53, 30, 0.1, 0.2, True, 0),
             "d": (54, 4, 0.0, 0.1, True, 0), "e": (55, 500, 0.2, 0.2, True, 1)}  # (seed, n, outliers, noise px, is_iterative, coplanar)


def pnp_case(seed, n, out_frac, noise, coplanar, h=376, w=1241):
    """seeded 3D-2D inputs of the PnP-tracker fixtures (also imported by the tests): view-1 keypoints on the pixel grid, their
    depth map (2 % holes, depths beyond the 50 m cap), view-2 keypoints of the moved camera + noise + uniform outliers,
    some of them outside the image"""
    r = np.random.Generator(np.random.PCG64(int(seed)))
    f = 718.856 * w / 1241.0
    K = np.array([[f, 0, 607.19 * w / 1241.0], [0, f, 185.22 * h / 376.0], [0, 0, 1]])
    kp1 = np.stack([r.integers(0, w, n), r.integers(0, h, n)], 1).astype(np.float64)
    Z = np.full(n, 20.0) if coplanar else r.uniform(2.0, 70.0, n)
    X = (np.linalg.inv(K) @ np.c_[kp1, np.ones(n)].T).T * Z[:, None]
    wv = np.array([0.002, 0.01, 0.001])
    th = np.linalg.norm(wv)
    k = wv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.02, 0.01, 0.8])
    x2 = (K @ ((R @ X.T).T + t).T).T
    kp2 = x2[:, :2] / x2[:, 2:] + r.normal(0, noise, (n, 2))
    o = r.random(n) < out_frac
    kp2[o] = np.stack([r.uniform(-50, w + 50, int(o.sum())), r.uniform(-30, h + 30, int(o.sum()))], 1)
    depth = np.zeros((h, w))
    depth[kp1[:, 1].astype(int), kp1[:, 0].astype(int)] = Z
    holes = r.random(n) < 0.02
    depth[kp1[holes, 1].astype(int), kp1[holes, 0].astype(int)] = 0.0
    return dict(kp1=np.ascontiguousarray(kp1), kp2=np.ascontiguousarray(kp2), depth_1=depth, K=K)
