"""CPU: the torch restatement of the nets (oracle/nets_torch.py) against fixtures produced by the
reference's own classes (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import nets_torch as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_liteflownet_matches_reference_fixture():
    g = np.load(os.path.join(G, "liteflownet_64x96.npz"))
    sd = O.liteflownet_state_dict(4869)
    with torch.no_grad():
        flows = O.liteflownet_forward(sd, torch.from_numpy(g["first"]), torch.from_numpy(g["second"]))
    for k in range(1, 6):
        want = g["flow%d" % k]
        got = flows[k].numpy()
        assert got.shape == want.shape
        # same torch build on the same ISA reproduces bit-for-bit; allow fp32 noise across CPUs
        assert np.abs(got - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), k


def test_monodepth2_matches_reference_fixture():
    g = np.load(os.path.join(G, "monodepth2_64x96.npz"))
    sd = O.monodepth2_state_dict(4869)
    with torch.no_grad():
        feats = O.resnet18_encoder(sd, torch.from_numpy(g["img"]))
        disps = O.depth_decoder(sd, feats)
    for i in range(5):
        assert np.abs(feats[i].numpy() - g["feat%d" % i]).max() <= 1e-4 * max(1.0, np.abs(g["feat%d" % i]).max())
    for s in range(4):
        assert np.abs(disps[s].numpy() - g["disp%d" % s]).max() <= 1e-5


def test_correlation_orders_agree():
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1, 64, 6, 9, generator=g)
    b = torch.randn(1, 64, 6, 9, generator=g)
    for s in (1, 2):
        x = O.correlation(a, b, s)
        y = O.correlation_cuda_order(a, b, s)
        assert x.shape == y.shape
        assert (x - y).abs().max() < 1e-5


def test_target_size():
    """DeepFlow.get_target_size as the reference evaluates it (fixture: the reference's own method, make_golden.py
    target_size): the oracle, the host-side helper and the library's C entry point all reproduce it"""
    import ctypes as C
    import importlib
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "target_size.npz"))["cases"]
    capi = importlib.import_module("df-vo_amd.capi")
    syn = importlib.import_module("df-vo_amd.synthetic")
    lib = capi.lib()
    assert tuple(fx[list(map(tuple, fx[:, :2])).index((376, 1241))][2:]) == (352, 1216)  # KITTI: NOT 384 x 1248
    assert tuple(fx[list(map(tuple, fx[:, :2])).index((192, 640))][2:]) == (224, 672)   # float64 rounding tips it up
    for h, w, th, tw in fx:
        assert O.get_target_size(int(h), int(w)) == (th, tw)
        assert syn._net_size(int(h), int(w)) == (th, tw)
        a, b = C.c_int(), C.c_int()
        capi.check(lib.dfvo_flow_target_size(int(h), int(w), C.byref(a), C.byref(b)))
        assert (a.value, b.value) == (th, tw)


def test_float64_anchor_is_the_same_function():
    """dtype=float64 (the accuracy anchor of tests/test_nets_gpu.py) evaluates the SAME function as the fp32 oracle: same
    inputs, weights and grid constants widened, so the two agree to fp32 rounding noise, the double run reproduces itself,
    and the fp32 run is bit-identical with and without the option (the pinned path is untouched)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from synth import image_pair
    sd = O.liteflownet_state_dict(4869)
    a, b = image_pair(70, 100, seed=1071)
    O._grid_cache.clear()
    f32 = O.flow_inference(sd, a, b)
    f32b = O.flow_inference(sd, a, b, dtype=torch.float32)
    f64 = O.flow_inference(sd, a, b, dtype=torch.float64)
    f64b = O.flow_inference(sd, a, b, dtype=torch.float64)
    for x, y, u, v in zip(f32, f32b, f64, f64b):
        assert x.dtype == np.float32 and u.dtype == np.float64
        assert np.array_equal(x, y) and np.array_equal(u, v)
        e = np.abs(x.astype(np.float64) - u)
        assert 0 < e.max() <= 2e-3 and np.median(e) <= 1e-4, (e.max(), np.median(e))
    dsd = O.monodepth2_state_dict(4869)
    img = image_pair(64, 96, seed=55)[0]
    d32, d64 = O.depth_inference(dsd, img), O.depth_inference(dsd, img, dtype=torch.float64)
    assert d32.dtype == np.float32 and d64.dtype == np.float64
    assert np.abs(d32 - d64).max() <= 1e-3 * np.abs(d64).max()
