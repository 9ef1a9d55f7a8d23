"""Seeded synthetic inputs: network weights with the reference's own
initialisation, textured image pairs, two-view correspondences and dense rigid-scene flow/depth maps.
Shared by bench.py, __graft_entry__.smoke(), the tests and (weights only) the oracle.  No network, no
datasets, no checkpoints are available in the build environment, so every measurement uses these."""
import math

import numpy as np
import torch

KER = [0, 0, 7, 5, 5, 3, 3]

def liteflownet_state_dict(seed=4869, gain=1.0):
    """Kaiming-normal conv weights, zero bias (lite_flow_net.py:273-282); `gain` < 1 tames activations."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, kh, kw, bias=True):
        fan_in = cin * kh * kw
        std = math.sqrt(2.0) / math.sqrt(fan_in)
        sd[name + '.weight'] = torch.randn(cout, cin, kh, kw, generator=g) * std * gain
        if bias:
            sd[name + '.bias'] = torch.randn(cout, generator=g) * 0.01

    p = 'moduleFeatures.'
    conv(p + 'moduleOne.0', 32, 3, 7, 7)
    conv(p + 'moduleTwo.0', 32, 32, 3, 3)
    conv(p + 'moduleTwo.2', 32, 32, 3, 3)
    conv(p + 'moduleTwo.4', 32, 32, 3, 3)
    conv(p + 'moduleThr.0', 64, 32, 3, 3)
    conv(p + 'moduleThr.2', 64, 64, 3, 3)
    conv(p + 'moduleFou.0', 96, 64, 3, 3)
    conv(p + 'moduleFou.2', 96, 96, 3, 3)
    conv(p + 'moduleFiv.0', 128, 96, 3, 3)
    conv(p + 'moduleSix.0', 192, 128, 3, 3)
    featc = [0, 32, 32, 64, 96, 128, 192]
    for lvl in [2, 3, 4, 5, 6]:
        k = KER[lvl]
        cm = 64 if lvl == 2 else featc[lvl]
        pm = 'moduleMatching.%d.' % (lvl - 2)
        ps = 'moduleSubpixel.%d.' % (lvl - 2)
        pr = 'moduleRegularization.%d.' % (lvl - 2)
        if lvl == 2:
            conv(pm + 'moduleFeat.0', 64, 32, 1, 1)
            conv(ps + 'moduleFeat.0', 64, 32, 1, 1)
        if lvl != 6:
            # ConvTranspose2d(2, 2, 4, groups=2): weight [2, 1, 4, 4]; near-bilinear init keeps flows sane
            sd[pm + 'moduleUpflow.weight'] = (torch.randn(2, 1, 4, 4, generator=g) * 0.05 + 0.25) * gain
        if lvl < 4:
            sd[pm + 'moduleUpcorr.weight'] = (torch.randn(49, 1, 4, 4, generator=g) * 0.05 + 0.25)
        conv(pm + 'moduleMain.0', 128, 49, 3, 3)
        conv(pm + 'moduleMain.2', 64, 128, 3, 3)
        conv(pm + 'moduleMain.4', 32, 64, 3, 3)
        conv(pm + 'moduleMain.6', 2, 32, k, k)
        conv(ps + 'moduleMain.0', 128, 2 * cm + 2, 3, 3)
        conv(ps + 'moduleMain.2', 64, 128, 3, 3)
        conv(ps + 'moduleMain.4', 32, 64, 3, 3)
        conv(ps + 'moduleMain.6', 2, 32, k, k)
        if lvl < 5:
            conv(pr + 'moduleFeat.0', 128, featc[lvl], 1, 1)
        conv(pr + 'moduleMain.0', 128, 131 if lvl < 6 else 195, 3, 3)
        conv(pr + 'moduleMain.2', 128, 128, 3, 3)
        conv(pr + 'moduleMain.4', 64, 128, 3, 3)
        conv(pr + 'moduleMain.6', 64, 64, 3, 3)
        conv(pr + 'moduleMain.8', 32, 64, 3, 3)
        conv(pr + 'moduleMain.10', 32, 32, 3, 3)
        if lvl >= 5:
            conv(pr + 'moduleDist.0', k * k, 32, k, k)
        else:
            conv(pr + 'moduleDist.0', k * k, 32, k, 1)
            conv(pr + 'moduleDist.1', k * k, k * k, 1, k)
        conv(pr + 'moduleScaleX', 1, k * k, 1, 1)
        conv(pr + 'moduleScaleY', 1, k * k, 1, 1)
    # flow heads: scale so that per-level flows are of the order of a pixel (warps get exercised)
    for key in list(sd.keys()):
        if key.endswith('moduleMain.6.weight'):
            sd[key] = sd[key] * 3.0
    return sd


def monodepth2_state_dict(seed=4869):
    """ResNet18 encoder (kaiming fan_out, BN identity-ish with seeded statistics) + decoder."""
    g = torch.Generator().manual_seed(seed + 1)
    sd = {}

    def conv(name, cout, cin, k, bias):
        std = math.sqrt(2.0 / (cin * k * k))
        sd[name + '.weight'] = torch.randn(cout, cin, k, k, generator=g) * std
        if bias:
            sd[name + '.bias'] = torch.randn(cout, generator=g) * 0.01

    def bn(name, c):
        sd[name + '.weight'] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + '.bias'] = 0.05 * torch.randn(c, generator=g)
        sd[name + '.running_mean'] = 0.05 * torch.randn(c, generator=g)
        sd[name + '.running_var'] = 1.0 + 0.1 * torch.rand(c, generator=g)

    conv('encoder.conv1', 64, 3, 7, False)
    bn('encoder.bn1', 64)
    ch = [64, 64, 128, 256, 512]
    for li in range(1, 5):
        cin, cout = ch[li - 1], ch[li]
        for b in range(2):
            p = 'encoder.layer%d.%d.' % (li, b)
            conv(p + 'conv1', cout, cin if b == 0 else cout, 3, False)
            bn(p + 'bn1', cout)
            conv(p + 'conv2', cout, cout, 3, False)
            bn(p + 'bn2', cout)
            if li > 1 and b == 0:
                conv(p + 'downsample.0', cout, cin, 1, False)
                bn(p + 'downsample.1', cout)
    dec = [16, 32, 64, 128, 256]
    for i in range(4, -1, -1):
        idx0 = (4 - i) * 2
        cin0 = 512 if i == 4 else dec[i + 1]
        conv('decoder.%d.conv.conv' % idx0, dec[i], cin0, 3, True)
        conv('decoder.%d.conv.conv' % (idx0 + 1), dec[i], dec[i] + (ch[i - 1] if i > 0 else 0), 3, True)
    for s in range(4):
        conv('decoder.%d.conv' % (10 + s), 1, dec[s], 3, True)
    return sd


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


def rigid_scene(h, w, seed=1, T=None, noise_px=0.05, bad_frac=0.35):
    """Dense forward flow / consistency map / current-view depth of a static ramp scene seen from a
    moving camera: depth d(y) = max(5, 80 (1 - y/H)), KITTI-like intrinsics.
    Returns dict(K, flow [2,h,w] f32, diff [h,w] f32, depth_cur / depth_ref [h,w] f64 (already cropped/capped), R, t)."""
    from scipy import ndimage
    rng = np.random.Generator(np.random.PCG64(seed))
    f = 718.856 * w / 1241.0
    K = np.array([[f, 0, 607.19 * w / 1241.0], [0, f, 185.22 * h / 376.0], [0, 0, 1]])
    if T is None:
        wv = np.array([0.002, 0.01, 0.001])
        tv = np.array([0.02, 0.01, 0.8])
    else:
        wv, tv = T
    th = np.linalg.norm(wv)
    k = wv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d = np.maximum(5.0, 80.0 * (1.0 - yy / h)) * (1.0 + 0.05 * np.sin(xx / 37.0))
    rays = np.linalg.inv(K) @ np.stack([xx.ravel(), yy.ravel(), np.ones(h * w)])
    X = rays * d.ravel()
    X2 = R @ X + tv[:, None]
    p2 = K @ X2
    u2, v2 = p2[0] / p2[2], p2[1] / p2[2]
    flow = np.stack([(u2 - xx.ravel()).reshape(h, w), (v2 - yy.ravel()).reshape(h, w)])
    flow = flow + rng.normal(0, noise_px, flow.shape)
    # forward-backward inconsistency: mostly tiny, a fraction of the image unreliable
    diff = np.abs(rng.normal(0, 0.03, (h, w)))
    bad = ndimage.gaussian_filter(rng.standard_normal((h, w)), 6) > np.quantile(
        ndimage.gaussian_filter(rng.standard_normal((h, w)), 6), 1 - bad_frac)
    diff[bad] += rng.uniform(0.2, 3.0, int(bad.sum()))
    # corrupt the flow where it is flagged unreliable (so that masked-out pixels really are outliers)
    flow[:, bad] += rng.normal(0, 3.0, (2, int(bad.sum())))
    # current-view depth: splat z' to the nearest pixel, fill holes by nearest neighbour
    dc = np.zeros((h, w))
    iu, iv = np.rint(u2).astype(int), np.rint(v2).astype(int)
    ok = (iu >= 0) & (iu < w) & (iv >= 0) & (iv < h)
    dc[iv[ok], iu[ok]] = X2[2][ok]
    hole = dc == 0
    idx = ndimage.distance_transform_edt(hole, return_distances=False, return_indices=True)
    dc = dc[tuple(idx)]
    dc = dc * (1.0 + rng.normal(0, 0.02, dc.shape))  # CNN depth noise
    scale_true = 1.0  # depth map is metric for this scene, E translation has unit norm -> scale = |t|
    y0 = int(h * 0.3)
    proc = dc.copy()
    proc[:y0] = 0
    proc[~((proc < 50) & (proc > 0))] = 0
    # reference-view depth (same noise model, same crop / cap), drawn last so that the other maps keep their values
    dr = d * (1.0 + rng.normal(0, 0.02, d.shape))
    proc_ref = dr.copy()
    proc_ref[:y0] = 0
    proc_ref[~((proc_ref < 50) & (proc_ref > 0))] = 0
    return dict(K=K, flow=flow.astype(np.float32), diff=diff.astype(np.float32), depth_cur=proc, depth_ref=proc_ref, R=R,
                t=tv, scale_true=scale_true)
