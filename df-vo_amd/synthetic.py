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


# ==================================================================================================================
# Coded tunnel world: frames that CARRY their own optical flow and depth, plus network weights that read them out.
#
# Random-weight networks give incoherent flow, so with them the nets -> keypoints -> RANSAC data path of
# libs/dfvo.py:299-345,121-262 can only be fed from outside.  No trained weights / KITTI frames exist in the build
# environment (no network), hence this construction: a convex rectangular tunnel (closed-form ray casting, no
# occlusions, forward and backward flow exactly consistent) is rendered into uint8 frames whose colour channels hold
# dithered CODES of the flow and depth fields, and a handful of channels of otherwise random LiteFlowNet / monodepth2
# weights are overwritten so that the real layer stack (7x7 conv, stride-2 conv, 1x1 conv, the Subpixel stack with its
# LeakyReLUs, the Regularization module's softmax-weighted local average; 7x7 stride-2 conv, BatchNorm, skip
# connection, ELU decoder, sigmoid head) decodes them.  Every kernel of both nets still runs at full cost on dense
# random weights; only the flow / disparity heads listen to the decoding channels.  Used by the end-to-end tests, the
# trajectory test and `bench.py --solver-inputs nets`; the torch-CPU oracle runs the same weights on the same frames.
#
# Two frame encodings:
#   "mux"   (image size == flow-net size, i.e. both divisible by 32; e.g. the reference's default 192x640): the 2x2
#           pixel block under a level-2 feature pixel holds  (R,G)@(0,0) = forward flow to the next frame,
#           (R,G)@(1,1) = backward flow to the previous frame,  R@(0,1) = frame counter;  B of every pixel = depth.
#           The Subpixel stack gates "first.fwd if second is the later frame else first.bwd" with exact ReLUs built
#           from LeakyReLU pairs, so fwd AND bwd flow are the true fields of an arbitrarily long sequence.
#   "pot"   (any size; the frame goes through the net's bilinear input resize, which spatial multiplexing does not
#           survive): (R,G) hold a potential x_k with x_k - x_{k+1} = fwd_k / (2 range); the net outputs
#           2 range (x_first - x_second), i.e. bwd_k(p) = -fwd_k(p) at the same pixel.  x accumulates, so this
#           encoding is for short sequences with small motion (tests: 4 frames; bench: one pair).
# ==================================================================================================================
DEPTH_LOGIT = (-5.0, -1.7)  # disparity logit at B = 0 / 255  ->  depth 70.2 m ... 3.48 m through monodepth2's head


def kitti_K(h, w):
    f = 718.856 * w / 1241.0
    return np.array([[f, 0, 607.19 * w / 1241.0], [0, f, 185.22 * h / 376.0], [0, 0, 1]])


def _depth_code_table():
    """depth [m] that monodepth2's head (sigmoid -> 1/(0.01 + 9.99 s) * 5.4) yields for each 8-bit code"""
    code = np.arange(256, dtype=np.float64)
    logit = DEPTH_LOGIT[0] + (DEPTH_LOGIT[1] - DEPTH_LOGIT[0]) * code / 255.0
    s = 1.0 / (1.0 + np.exp(-logit))
    return 5.4 / (0.01 + 9.99 * s)  # strictly decreasing in the code


def encode_depth(z):
    """nearest 8-bit code for a depth map (comparisons against the 256-entry table only)"""
    tab = _depth_code_table()
    mid = 0.5 * (tab[1:] + tab[:-1])  # decreasing
    return (255 - np.searchsorted(mid[::-1], z, side="left")).astype(np.uint8)


def crafted_monodepth2_state_dict(seed=4869):
    """random ResNet18 + decoder whose disparity head reads the blue channel:  logit = L0 + (L1 - L0) B/255.
    Path: conv1 centre tap (+B, -B) -> BN identity -> ReLU pair -> skip connection into upconv(1,1): (p - n) + 3 > 0
    (ELU = identity) -> upconv(0,0), nearest x2, upconv(0,1) centre taps -> dispconv -> sigmoid."""
    sd = monodepth2_state_dict(seed)
    w = sd['encoder.conv1.weight']
    w[0:2] = 0
    w[0, 2, 3, 3] = 1.0
    w[1, 2, 3, 3] = -1.0
    for k, v in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0)):
        sd['encoder.bn1.' + k][0:2] = v
    w = sd['decoder.7.conv.conv.weight']  # upconv(1,1): cat(upsampled 32, features[0] 64) -> 32
    w[0] = 0
    w[0, 32 + 0, 1, 1] = 1.0
    w[0, 32 + 1, 1, 1] = -1.0
    sd['decoder.7.conv.conv.bias'][0] = 3.0
    for name in ('decoder.8.conv.conv', 'decoder.9.conv.conv'):  # upconv(0,0), upconv(0,1)
        w = sd[name + '.weight']
        w[0] = 0
        w[0, 0, 1, 1] = 1.0
        sd[name + '.bias'][0] = 0.0
    bn_scale = 1.0 / math.sqrt(1.0 + 1e-5)  # eval BatchNorm with unit variance
    a = (DEPTH_LOGIT[1] - DEPTH_LOGIT[0]) * 0.225 / bn_scale
    b = DEPTH_LOGIT[0] + (DEPTH_LOGIT[1] - DEPTH_LOGIT[0]) * 0.45
    w = sd['decoder.10.conv.weight']
    w[:] = 0
    w[0, 0, 1, 1] = a
    sd['decoder.10.conv.bias'][0] = b - 3.0 * a
    return sd


def crafted_liteflownet_state_dict(h, w, mode, flow_range=32.0, seed=4869):
    """random LiteFlowNet whose level-2 flow reads the frame codes (see the block comment above).
    (h, w) = image size (the output scaling of deep_flow.py:106-129 is folded into the head gain)."""
    assert mode in ("mux", "pot")
    sd = liteflownet_state_dict(seed)
    nh, nw = _net_size(h, w)
    if mode == "mux":
        assert (nh, nw) == (h, w), "the multiplexed encoding needs image size == flow-net size"
    ncode = 5 if mode == "mux" else 2

    def passthrough(name, n, tap):
        wt, b = sd[name + '.weight'], sd[name + '.bias']
        wt[:n] = 0
        b[:n] = 0
        for c in range(n):
            wt[c, c, tap, tap] = 1.0

    p = 'moduleFeatures.'
    passthrough(p + 'moduleOne.0', 2, 3)  # R, G at the centre tap of the 7x7
    wt, b = sd[p + 'moduleTwo.0.weight'], sd[p + 'moduleTwo.0.bias']  # 3x3 stride 2 pad 1: tap (1,1) = pixel (2i,2j)
    wt[:ncode] = 0
    b[:ncode] = 0
    wt[0, 0, 1, 1] = 1.0
    wt[1, 1, 1, 1] = 1.0
    if mode == "mux":
        wt[2, 0, 2, 2] = 1.0  # R @ (2i+1, 2j+1): backward u
        wt[3, 1, 2, 2] = 1.0  # G @ (2i+1, 2j+1): backward v
        wt[4, 0, 1, 2] = 1.0  # R @ (2i, 2j+1): frame counter
    passthrough(p + 'moduleTwo.2', ncode, 1)
    passthrough(p + 'moduleTwo.4', ncode, 1)
    passthrough('moduleSubpixel.0.moduleFeat.0', ncode, 0)
    # level-2 Matching contributes nothing: flow entering Subpixel is exactly 0 (second features warped by zero flow)
    sd['moduleMatching.0.moduleUpflow.weight'][:] = 0
    sd['moduleMatching.0.moduleMain.6.weight'][:] = 0
    sd['moduleMatching.0.moduleMain.6.bias'][:] = 0
    ps = 'moduleSubpixel.0.moduleMain.'
    F, S = 0, 64  # channel offsets of first / second features in cat([first, second, flow])
    w0, b0 = sd[ps + '0.weight'], sd[ps + '0.bias']
    w2, b2 = sd[ps + '2.weight'], sd[ps + '2.bias']
    w4, b4 = sd[ps + '4.weight'], sd[ps + '4.bias']
    w6, b6 = sd[ps + '6.weight'], sd[ps + '6.bias']
    r = 1.0 / 0.99  # relu(z) = (leaky(z) + 0.1 leaky(-z)) / 0.99 for slope 0.1
    if mode == "mux":
        w0[:6] = 0
        b0[:6] = 0
        for c in range(4):
            w0[c, F + c, 1, 1] = 1.0  # a_u, a_v (first.fwd), b_u, b_v (first.bwd)
        w0[4, S + 4, 1, 1], w0[4, F + 4, 1, 1] = 255.0, -255.0  # d = +1: second is the later frame
        w0[5, S + 4, 1, 1], w0[5, F + 4, 1, 1] = -255.0, 255.0
        w2[:8] = 0
        b2[:8] = 0
        for comp in range(2):  # s = relu(d);  z1 = a + s - 1 (selected when s = 1),  z2 = b - s (when s = 0)
            o = 4 * comp
            for sign, row in ((1.0, o), (-1.0, o + 1)):
                w2[row, comp, 1, 1] = sign
                w2[row, 4, 1, 1], w2[row, 5, 1, 1] = sign * r, sign * 0.1 * r
                b2[row] = -sign
            for sign, row in ((1.0, o + 2), (-1.0, o + 3)):
                w2[row, 2 + comp, 1, 1] = sign
                w2[row, 4, 1, 1], w2[row, 5, 1, 1] = -sign * r, -sign * 0.1 * r
        w4[:2] = 0
        b4[:2] = 0
        for comp in range(2):  # selected code = relu(z1) + relu(z2) in [0, 1]
            o = 4 * comp
            w4[comp, o, 1, 1], w4[comp, o + 1, 1, 1] = r, 0.1 * r
            w4[comp, o + 2, 1, 1], w4[comp, o + 3, 1, 1] = r, 0.1 * r
    else:
        w0[:4] = 0
        b0[:4] = 0
        w0[0, F + 0, 1, 1] = w0[1, F + 1, 1, 1] = w0[2, S + 0, 1, 1] = w0[3, S + 1, 1, 1] = 1.0
        for wt, b in ((w2, b2), (w4, b4)):
            wt[:4] = 0
            b[:4] = 0
            for c in range(4):
                wt[c, c, 1, 1] = 1.0
    # head: image-resolution flow [px] = 10 * (W_img / W_level2) * net output  (lite_flow_net.py:322-324, deep_flow.py:106-129)
    gu = 2.0 * flow_range / (10.0 * w / (nw // 2))
    gv = 2.0 * flow_range / (10.0 * h / (nh // 2))
    w6[:] = 0
    if mode == "mux":
        w6[0, 0, 3, 3], w6[1, 1, 3, 3] = gu, gv
        b6[0], b6[1] = -0.5 * gu, -0.5 * gv
    else:
        w6[0, 0, 3, 3], w6[0, 2, 3, 3] = gu, -gu
        w6[1, 1, 3, 3], w6[1, 3, 3, 3] = gv, -gv
        b6[:] = 0
    # Regularization: a convex combination of the 7x7 neighbourhood (ScaleX/Y = 1), near-uniform softmax weights
    pr = 'moduleRegularization.0.'
    for k in (pr + 'moduleScaleX', pr + 'moduleScaleY'):
        sd[k + '.weight'][:] = 1.0
        sd[k + '.bias'][:] = 0.0
    sd[pr + 'moduleDist.1.weight'] *= 0.02
    sd[pr + 'moduleDist.1.bias'] *= 0.02
    return sd


def _net_size(h, w):
    """flow-net input size as the reference's DeepFlow.get_target_size (deep_flow.py:89-105) really computes it: the
    function rebinds h, w to the candidate arrays before comparing aspect ratios, so the compared matrix is
    |h_i * (1/w_j) - h_j / w_j| (float64, first minimum in row-major order): (floor, floor) multiples of 32 unless rounding
    makes entry [0][0] non-zero -- 376x1241 -> 352x1216, 192x640 -> 224x672, 256x640 / 384x1248 unchanged"""
    hs = [32.0 * (h // 32), 32.0 * (h // 32 + 1)]
    ws = [32.0 * (w // 32), 32.0 * (w // 32 + 1)]
    best, arg = None, None
    for i in range(4):
        rr = abs(hs[i // 2] * (1.0 / ws[i % 2]) - hs[i % 2] / ws[i % 2])
        if best is None or rr < best:
            best, arg = rr, i
    return int(hs[arg // 2]), int(ws[arg % 2])


def tunnel_poses(n, step=1.0, seed=7):
    """camera-to-world poses [n,4,4] (KITTI convention: x right, y down, z forward): forward drive with gentle yaw /
    pitch oscillation and lateral sway, first pose = identity"""
    k = np.arange(n, dtype=np.float64)
    yaw = 0.06 * np.sin(k / 11.0) + 0.004 * k / max(n, 1)
    pitch = 0.008 * np.sin(k / 7.0 + 1.0)
    x = 1.2 * (np.cos(k / 17.0) - 1.0)
    y = 0.08 * np.sin(k / 5.0)
    z = step * k + 0.15 * np.sin(k / 3.0)
    yaw, pitch = yaw - yaw[0], pitch - pitch[0]
    y, z = y - y[0], z - z[0]
    T = np.zeros((n, 4, 4))
    for i in range(n):
        cy_, sy_ = math.cos(yaw[i]), math.sin(yaw[i])
        cp, sp = math.cos(pitch[i]), math.sin(pitch[i])
        Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        T[i, :3, :3] = Ry @ Rx
        T[i, :3, 3] = [x[i], y[i], z[i]]
        T[i, 3, 3] = 1
    return T


def tunnel_poses_lateral(n, step=0.4):
    """sideways drive (x by `step`, z by step / 4 per frame, slight yaw): on the tunnel's floor and ceiling the depth does not
    change along the flow direction, so forward and backward flow agree at the SAME pixel -- the motion for which the
    "pot" encoding (bwd(p) = -fwd(p)) passes the consistency check where parallax is large, and the E-tracker is accepted"""
    T = np.tile(np.eye(4), (n, 1, 1))
    for k in range(n):
        yaw = 0.004 * k
        c, s_ = math.cos(yaw), math.sin(yaw)
        T[k, :3, :3] = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
        T[k, :3, 3] = [step * k, 0.02 * k, 0.25 * step * k]
    return T


def tunnel_cast(K, T_wc, py, px, half_width=15.0, cam_height=5.0, ceiling=12.0):
    """world points hit by the pixel rays (py, px arrays) of camera T_wc in the tunnel |X| <= half_width,
    -ceiling <= Y <= cam_height (infinite along Z; convex, so every point is visible from every interior camera).
    Returns (Xw [3, ...], depth = z in the camera frame)."""
    Kinv = np.linalg.inv(K)
    d_c = np.stack([Kinv[0, 0] * px + Kinv[0, 2], Kinv[1, 1] * py + Kinv[1, 2], np.ones_like(px)])  # z = 1 rays
    R, o = T_wc[:3, :3], T_wc[:3, 3]
    d_w = np.tensordot(R, d_c, axes=(1, 0))
    tbest = np.full(px.shape, np.inf)
    for axis, plane in ((0, half_width), (0, -half_width), (1, cam_height), (1, -ceiling)):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (plane - o[axis]) / d_w[axis]
        t = np.where((t > 0) & np.isfinite(t), t, np.inf)
        tbest = np.minimum(tbest, t)
    tbest = np.minimum(tbest, 1e4)  # rays along the tunnel axis
    Xw = o.reshape(3, *([1] * px.ndim)) + d_w * tbest
    return Xw, tbest


def tunnel_flow(K, T_wc_from, T_wc_to, py, px, **world):
    """flow (u, v) of the pixels (py, px) of camera `from` into camera `to`, and their depth in `from`"""
    Xw, z = tunnel_cast(K, T_wc_from, py, px, **world)
    R, o = T_wc_to[:3, :3], T_wc_to[:3, 3]
    Xc = np.tensordot(R.T, Xw - o.reshape(3, *([1] * px.ndim)), axes=(1, 0))
    u = K[0, 0] * Xc[0] / Xc[2] + K[0, 2]
    v = K[1, 1] * Xc[1] / Xc[2] + K[1, 2]
    return u - px, v - py, z


def coded_tunnel_sequence(h, w, n_frames, mode="mux", flow_range=32.0, step=1.0, seed=7, poses=None, world=None):
    """uint8 frames [n,h,w,3] of the coded tunnel world + ground truth.
    Returns dict(frames, poses [n,4,4] camera-to-world, K, mode, flow_range, world)."""
    world = dict(world or {})
    K = kitti_K(h, w)
    T = tunnel_poses(n_frames, step, seed) if poses is None else np.asarray(poses, np.float64)
    rng = np.random.Generator(np.random.PCG64(seed))
    nh, nw = _net_size(h, w)
    frames = np.zeros((n_frames, h, w, 3), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)

    def code(v):  # dithered 8-bit code of a value in [0, 1]
        return np.clip(np.floor(v * 255.0 + rng.random(v.shape)), 0, 255).astype(np.uint8)

    if mode == "mux":
        assert (nh, nw) == (h, w) and n_frames <= 256
        h2, w2 = h // 2, w // 2
        # image position a level-2 pixel stands for once the level-2 flow is resized to the image (align_corners)
        py = (np.arange(h2, dtype=np.float64) * (h - 1) / (h2 - 1))[:, None] * np.ones((1, w2))
        px = np.ones((h2, 1)) * (np.arange(w2, dtype=np.float64) * (w - 1) / (w2 - 1))[None, :]
    else:
        x_pot = np.full((2, h, w), 0.5)
    for k in range(n_frames):
        _, z = tunnel_cast(K, T[k], yy, xx, **world)
        frames[k, :, :, 2] = encode_depth(z)
        if mode == "mux":
            frames[k, :, :, 0:2] = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)  # filler texture
            if k + 1 < n_frames:
                u, v, _ = tunnel_flow(K, T[k], T[k + 1], py, px, **world)
            else:
                u = v = np.zeros((h2, w2))
            frames[k, 0::2, 0::2, 0] = code(0.5 + u / (2 * flow_range))
            frames[k, 0::2, 0::2, 1] = code(0.5 + v / (2 * flow_range))
            if k > 0:
                u, v, _ = tunnel_flow(K, T[k], T[k - 1], py, px, **world)
            else:
                u = v = np.zeros((h2, w2))
            frames[k, 1::2, 1::2, 0] = code(0.5 + u / (2 * flow_range))
            frames[k, 1::2, 1::2, 1] = code(0.5 + v / (2 * flow_range))
            frames[k, 0::2, 1::2, 0] = k
        else:
            frames[k, :, :, 0] = code(x_pot[0])
            frames[k, :, :, 1] = code(x_pot[1])
            if k + 1 < n_frames:
                u, v, _ = tunnel_flow(K, T[k], T[k + 1], yy, xx, **world)
                x_pot[0] -= u / (2 * flow_range)
                x_pot[1] -= v / (2 * flow_range)
    return dict(frames=frames, poses=T, K=K, mode=mode, flow_range=flow_range, world=world)


def tunnel_truth(seq, k):
    """ground-truth forward flow [2,h,w] (frame k -> k+1), backward flow (k+1 -> k) and depth of frame k+1, full resolution"""
    fr = seq["frames"]
    h, w = fr.shape[1:3]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    u, v, _ = tunnel_flow(seq["K"], seq["poses"][k], seq["poses"][k + 1], yy, xx, **seq["world"])
    ub, vb, z1 = tunnel_flow(seq["K"], seq["poses"][k + 1], seq["poses"][k], yy, xx, **seq["world"])
    return np.stack([u, v]), np.stack([ub, vb]), z1


def write_weight_files(dirname, fsd, dsd, feed_h=192, feed_w=640):
    """the on-disk formats the reference loads: `network-default.pytorch` = a bare LiteFlowNet state_dict
    (lite_flow.py:45-46); a directory with `encoder.pth` = ResnetEncoder state_dict (keys 'encoder.*', torchvision's unused
    fc layer and BatchNorm counters included, as a real checkpoint has them) plus the scalar entries 'height' / 'width' /
    'use_stereo' (monodepth2.py:47-50,70-71) and `depth.pth` = DepthDecoder state_dict (:54-57).
    Returns (flow weight path, depth weight directory)."""
    import os
    ddir = os.path.join(dirname, "depth")
    os.makedirs(ddir, exist_ok=True)
    flow_path = os.path.join(dirname, "network-default.pytorch")
    torch.save({k: v.clone() for k, v in fsd.items()}, flow_path)
    enc = {k: v.clone() for k, v in dsd.items() if k.startswith("encoder.")}
    for k in list(enc.keys()):
        if k.endswith(".running_mean"):
            enc[k[:-len("running_mean")] + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    enc["encoder.fc.weight"] = torch.zeros(1000, 512)
    enc["encoder.fc.bias"] = torch.zeros(1000)
    enc["height"], enc["width"], enc["use_stereo"] = feed_h, feed_w, True
    torch.save(enc, os.path.join(ddir, "encoder.pth"))
    torch.save({k: v.clone() for k, v in dsd.items() if k.startswith("decoder.")}, os.path.join(ddir, "depth.pth"))
    return flow_path, ddir
