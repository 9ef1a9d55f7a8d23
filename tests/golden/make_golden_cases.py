"""Seeded case generators of the fixtures that code WITHOUT torch must regenerate too (tests/second_env_ref_tracker.py runs
under an interpreter that has none); make_golden.py re-exports them."""
import numpy as np

from synth import two_view  # tests/synth.py (tests/ is on sys.path for every importer of this module)


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


def tracker_case(seed, n=2000, out_frac=0.3, noise=0.15, h=376, w=1241):
    """seeded inputs for the E-tracker fixtures (also imported by the tests)"""
    x1, x2, R, t, K, o = two_view(n, out_frac, noise, seed, w=w, h=h)
    # view 1 = reference frame, view 2 = current frame; CNN depth of the current view at int(kp_cur):
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    Kinv = np.linalg.inv(K)
    depth = np.zeros((h, w))
    zs = rng.uniform(5, 60, n)
    ix, iy = x2[:, 0].astype(int), x2[:, 1].astype(int)
    ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    depth[iy[ok], ix[ok]] = zs[ok]
    return dict(kp_ref=x1, kp_cur=x2, K=K, depth_cur=depth, R=R, t=t, outliers=o)


def variant_case(tag):
    """inputs of the e_tracker_variants fixtures (also imported by the tests): tracker_case with a CNN depth that agrees
    with the geometry (true depth of view 2 x 1.25, 2 % noise, a fifth of the pixels arbitrary); case 'p': a planar
    scene (keypoints related by a homography) -- the homography explains as many matches as the essential matrix"""
    seed, n, of, noise = {"a": (71, 2000, 0.3, 0.15), "b": (72, 1500, 0.5, 0.25), "p": (73, 2000, 0.2, 0.05),
                          "d": (74, 2000, 0.97, 0.2)}[tag]
    c = tracker_case(seed, n, of, noise)
    r = np.random.Generator(np.random.PCG64(seed))
    X = np.stack([r.uniform(-20, 20, n), r.uniform(-3, 3, n), r.uniform(5, 60, n)], 1)  # two_view's points
    z2 = ((c["R"] @ X.T).T + c["t"])[:, 2]
    g = np.random.Generator(np.random.PCG64(seed + 1000))
    zs = z2 * 1.25 * (1 + g.normal(0, 0.02, n))
    wild = g.random(n) < 0.2
    zs[wild] = g.uniform(5, 60, int(wild.sum()))
    h, w = c["depth_cur"].shape
    kp_ref, kp_cur = c["kp_ref"], c["kp_cur"]
    if tag == "p":
        Hm = np.array([[1.02, 0.01, 6.0], [-0.004, 1.015, 2.0], [1e-5, -2e-5, 1.0]])
        q = (Hm @ np.c_[kp_ref, np.ones(n)].T).T
        kp_cur = q[:, :2] / q[:, 2:] + g.normal(0, noise, (n, 2))
        kp_cur[c["outliers"]] = np.stack([g.uniform(0, w, int(c["outliers"].sum())),
                                          g.uniform(0, h, int(c["outliers"].sum()))], 1)
        kp_cur = np.ascontiguousarray(kp_cur)
    depth = np.zeros((h, w))
    ix, iy = kp_cur[:, 0].astype(int), kp_cur[:, 1].astype(int)
    ok = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    depth[iy[ok], ix[ok]] = zs[ok]
    return dict(seed=seed, kp_ref=kp_ref, kp_cur=kp_cur, K=c["K"], depth_cur=depth)
