"""GPU parity of the individual HIP operators against torch-CPU fp32 references of the same op
(each a one-line restatement of the reference call site named in include/dfvo_hip.h).
Tolerances are stated per test; fp32 MFMA is an exact fmaf chain, so conv errors are pure
summation-order noise."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets_torch as O
from util import nhwc_dev, nchw_host, ptr, report

pytestmark = pytest.mark.gpu


def run_conv(capi, x0, w, b, stride=1, pad=(0, 0), pad_mode=0, act=0, act_param=0.0, x1=None, up0=0, res=None,
             cs0=None, cs1=None):
    lib = capi.lib()
    n, c0 = x0.shape[0], x0.shape[1]
    c1 = x1.shape[1] if x1 is not None else 0
    cout, cin, kh, kw = w.shape
    assert cin == c0 + c1
    H, W = (x0.shape[2] * 2, x0.shape[3] * 2) if up0 else (x0.shape[2], x0.shape[3])
    Ho = (H + 2 * pad[0] - kh) // stride + 1
    Wo = (W + 2 * pad[1] - kw) // stride + 1
    d0 = nhwc_dev(x0, cs0)
    d1 = nhwc_dev(x1, cs1) if x1 is not None else None
    dcs = (cout + 3) // 4 * 4
    dst = torch.zeros(n, Ho, Wo, dcs, device="cuda")
    dres = nhwc_dev(res) if res is not None else None
    desc = capi.ConvDesc(N=n, H=H, W=W, kh=kh, kw=kw, stride=stride, pad_h=pad[0], pad_w=pad[1], pad_mode=pad_mode,
                         c0=c0, cs0=d0.shape[3], co0=0, up0=up0, c1=c1, cs1=d1.shape[3] if d1 is not None else 0, co1=0,
                         cout=cout, act=act, act_param=act_param, res_cs=dres.shape[3] if dres is not None else 0,
                         res_co=0, dst_cs=dcs, dst_co=0)
    wn = np.ascontiguousarray(w.numpy())
    bn = np.ascontiguousarray(b.numpy()) if b is not None else None
    capi.check(lib.dfvo_conv2d(C.byref(desc), ptr(d0), ptr(d1), capi.as_ptr(wn), capi.as_ptr(bn), ptr(dres), ptr(dst),
                               None))
    torch.cuda.synchronize()
    return nchw_host(dst, cout)


def ref_act(y, act, a):
    return {0: lambda t: t, 1: lambda t: F.leaky_relu(t, a), 2: F.relu, 3: lambda t: F.elu(t, a), 4: torch.sigmoid}[act](y)


CASES = [
    # name, N, H, W, c0, c1, cout, kh, kw, stride, pad, pad_mode, act, up0, res
    ("3x3_64_128_leaky", 2, 24, 40, 64, 0, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("7x7_3_32", 2, 40, 56, 3, 0, 32, 7, 7, 1, (3, 3), 0, 1, 0, False),
    ("3x3s2_32_64", 2, 48, 64, 32, 0, 64, 3, 3, 2, (1, 1), 0, 1, 0, False),
    ("1x1_32_64", 2, 33, 47, 32, 0, 64, 1, 1, 1, (0, 0), 0, 1, 0, False),
    ("cat_64+66_128", 2, 20, 30, 64, 66, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("cat_3+128_128", 2, 20, 30, 3, 128, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("head7x7_32_2_res", 2, 30, 44, 32, 0, 2, 7, 7, 1, (3, 3), 0, 0, 0, True),
    ("dist7x1_32_49", 2, 30, 44, 32, 0, 49, 7, 1, 1, (3, 0), 0, 0, 0, False),
    ("dist1x7_49_49", 2, 30, 44, 49, 0, 49, 1, 7, 1, (0, 3), 0, 0, 0, False),
    ("96_96", 2, 16, 24, 96, 0, 96, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("192_out", 2, 8, 12, 128, 0, 192, 3, 3, 2, (1, 1), 0, 1, 0, False),
    ("dec_refl_up_cat_elu", 1, 12, 20, 128, 128, 128, 3, 3, 1, (1, 1), 1, 3, 1, False),
    ("dec_refl_sigmoid_16_1", 1, 32, 48, 16, 0, 1, 3, 3, 1, (1, 1), 1, 4, 0, False),
    ("resnet_512_small", 1, 6, 20, 512, 0, 512, 3, 3, 1, (1, 1), 0, 2, 0, True),
    # small maps whose K range is divided over workgroups as well (conv_gemm_f16s.h, gridDim.z > 1) in f16x3 mode
    ("resnet_256_12x40", 1, 12, 40, 256, 0, 256, 3, 3, 1, (1, 1), 0, 2, 0, True),
    ("flow_L5_128_11x38", 2, 11, 38, 128, 0, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("flow_L6_192_6x19_cat", 2, 6, 19, 192, 52, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("big_M_128_128", 2, 96, 160, 128, 0, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("big_M_64_32", 2, 96, 160, 64, 0, 32, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("big_M_32_2_5x5", 2, 96, 160, 32, 0, 2, 5, 5, 1, (2, 2), 0, 0, 0, True),
    # shapes served by the LDS-window 3x3 kernel (M >= 30000): ragged tiles, two sources with a 4-channel tail,
    # a source that is not a multiple of 16 channels, reflection + x2 upsample + concat, residual
    ("win_ragged_128_128", 2, 99, 157, 128, 0, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("win_cat_128_4_to_128", 2, 96, 168, 128, 4, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("win_52_to_128", 2, 100, 152, 52, 0, 128, 3, 3, 1, (1, 1), 0, 1, 0, True),
    ("win_128_64", 2, 96, 160, 128, 0, 64, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("win_32_32_res", 2, 101, 163, 32, 0, 32, 3, 3, 1, (1, 1), 0, 1, 0, True),
    ("win_64_49", 2, 96, 160, 64, 0, 49, 3, 3, 1, (1, 1), 0, 0, 0, False),
    ("win_dec_refl_up_cat_elu", 1, 96, 160, 32, 64, 32, 3, 3, 1, (1, 1), 1, 3, 1, False),
    ("win_small_grid_128", 1, 120, 264, 128, 0, 128, 3, 3, 1, (1, 1), 0, 1, 0, False),
    ("win7_32_2_res", 2, 97, 170, 32, 0, 2, 7, 7, 1, (3, 3), 0, 0, 0, True),
    ("win5_32_2", 2, 48, 156, 32, 0, 2, 5, 5, 1, (2, 2), 0, 0, 0, False),
    # one- / two-channel heads on the direct kernel: two sources with a 4-channel tail, reflection padding, a source
    # that is not a multiple of 8 channels, ragged tiles; and a 5-channel 5x5 layer that stays on the window kernel
    ("head3_cat_32_4_to_2", 2, 45, 77, 32, 4, 2, 3, 3, 1, (1, 1), 0, 0, 0, True),
    ("head3_refl_sigmoid_64_1", 1, 96, 320, 64, 0, 1, 3, 3, 1, (1, 1), 1, 4, 0, False),
    ("head5_20_1", 2, 50, 70, 20, 0, 1, 5, 5, 1, (2, 2), 0, 1, 0, False),
    ("head7_up_refl_12_2", 1, 33, 50, 12, 0, 2, 7, 7, 1, (3, 3), 1, 0, 1, False),
    ("win5_32_5", 2, 48, 156, 32, 0, 5, 5, 5, 1, (2, 2), 0, 0, 0, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv(gpu, conv_precision, case):
    name, n, h, w, c0, c1, cout, kh, kw, stride, pad, pad_mode, act, up0, use_res = case
    g = torch.Generator().manual_seed(hash(name) % 10000)
    x0 = torch.randn(n, c0, h, w, generator=g)
    H, W = (2 * h, 2 * w) if up0 else (h, w)
    x1 = torch.randn(n, c1, H, W, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, kh, kw, generator=g) / np.sqrt((c0 + c1) * kh * kw)
    b = torch.randn(cout, generator=g) * 0.1
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up0 else x0
    if x1 is not None:
        xin = torch.cat([xin, x1], 1)
    if pad_mode == 1:
        ref = F.conv2d(F.pad(xin, (pad[1], pad[1], pad[0], pad[0]), mode="reflect"), wt, b, stride=stride)
    else:
        ref = F.conv2d(xin, wt, b, stride=stride, padding=pad)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res
    ref = ref_act(ref, act, 0.1 if act == 1 else 1.0)
    got = run_conv(gpu, x0, wt, b, stride, pad, pad_mode, act, 0.1 if act == 1 else 1.0, x1, up0, res)
    err, scale = report("conv " + name, got, ref)
    assert err <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("C,stride,h,w", [(64, 1, 12, 20), (64, 2, 24, 39), (96, 1, 12, 13), (128, 1, 9, 11),
                                           (192, 1, 6, 10), (32, 2, 17, 23)])
def test_correlation(gpu, C, stride, h, w):
    lib = gpu.lib()
    g = torch.Generator().manual_seed(C + stride)
    a = torch.randn(2, C, h, w, generator=g)
    b = torch.randn(2, C, h, w, generator=g)
    ho, wo = -(-h // stride), -(-w // stride)
    out = torch.zeros(2, ho, wo, 52, device="cuda")
    da, db = nhwc_dev(a), nhwc_dev(b)
    gpu.check(lib.dfvo_correlation(ptr(da), ptr(db), 2, h, w, C, stride, 1.0, ptr(out), None))
    torch.cuda.synchronize()
    got = nchw_host(out, 49)
    ref = O.correlation(a, b, stride)
    err, scale = report("correlation C=%d s=%d" % (C, stride), got, ref)
    assert err <= 1e-5 * max(1.0, scale)
    exact = O.correlation_cuda_order(a, b, stride)
    nbad = int((got != exact).sum())
    print("   bit-exact vs CUDA-order restatement: %d / %d differ" % (nbad, exact.numel()))
    assert nbad <= exact.numel() // 1000  # fmaf emulation via float64 can double-round on rare elements
    # leaky relu fused
    gpu.check(lib.dfvo_correlation(ptr(da), ptr(db), 2, h, w, C, stride, 0.1, ptr(out), None))
    torch.cuda.synchronize()
    assert (nchw_host(out, 49) - F.leaky_relu(got, 0.1)).abs().max() <= 1e-7


@pytest.mark.parametrize("C,h,w,mult", [(64, 24, 40, 2.5), (4, 48, 64, 10.0), (96, 7, 9, 0.625)])
def test_backward_warp(gpu, C, h, w, mult):
    lib = gpu.lib()
    g = torch.Generator().manual_seed(C)
    src = torch.randn(2, C, h, w, generator=g)
    flow = torch.randn(2, 2, h, w, generator=g) * 1.5
    flow[0, :, 0, 0] = torch.tensor([-30.0, 4.0])  # far out of range -> zeros
    O._grid_cache.clear()
    ref = O.backward_warp(src, flow * mult)
    lx = torch.linspace(-1.0, 1.0, w).numpy()
    ly = torch.linspace(-1.0, 1.0, h).numpy()
    dsrc = nhwc_dev(src)
    dflow = flow.permute(0, 2, 3, 1).contiguous().cuda()
    dst = torch.zeros(2, h, w, C, device="cuda")
    gpu.check(lib.dfvo_backward_warp(ptr(dsrc), ptr(dflow), mult, 2, h, w, C, gpu.as_ptr(lx), gpu.as_ptr(ly), ptr(dst),
                                     None))
    torch.cuda.synchronize()
    err, scale = report("warp C=%d" % C, nchw_host(dst, C), ref)
    assert err <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("C,cs,h,w", [(2, 4, 12, 20), (49, 52, 9, 14)])
def test_deconv_dw(gpu, C, cs, h, w):
    lib = gpu.lib()
    g = torch.Generator().manual_seed(C)
    x = torch.randn(2, C, h, w, generator=g)
    wt = torch.randn(C, 1, 4, 4, generator=g)
    ref = F.conv_transpose2d(x, wt, None, stride=2, padding=1, groups=C)
    dx = nhwc_dev(x, cs)
    dst = torch.zeros(2, 2 * h, 2 * w, cs, device="cuda")
    wn = np.ascontiguousarray(wt.numpy())
    gpu.check(lib.dfvo_deconv_dw4x4s2(ptr(dx), 2, h, w, C, cs, gpu.as_ptr(wn), ptr(dst), None))
    torch.cuda.synchronize()
    err, scale = report("deconv C=%d" % C, nchw_host(dst, C), ref)
    assert err <= 1e-5 * max(1.0, scale)


@pytest.mark.parametrize("ac,h,w,ho,wo", [(0, 48, 64, 24, 32), (1, 37, 53, 48, 64), (0, 24, 39, 12, 20), (1, 48, 64, 37, 53)])
def test_resize_bilinear(gpu, ac, h, w, ho, wo):
    lib = gpu.lib()
    g = torch.Generator().manual_seed(h)
    x = torch.randn(2, 4, h, w, generator=g)
    ref = F.interpolate(x, (ho, wo), mode="bilinear", align_corners=bool(ac))
    dx = nhwc_dev(x)
    dst = torch.zeros(2, ho, wo, 4, device="cuda")
    gpu.check(lib.dfvo_resize_bilinear(ptr(dx), 2, h, w, 4, ptr(dst), ho, wo, ac, None))
    torch.cuda.synchronize()
    err, scale = report("resize ac=%d" % ac, nchw_host(dst, 4), ref)
    assert err <= 1e-5 * max(1.0, scale)


def test_split_precision_modes(gpu):
    """conv modes (DFVO_CONV_PRECISION): exact fp32 (library default) and f16x3 (two f16 planes per operand, three exact
    products per term: fp32-class accuracy) on the same 132 -> 64 window layer.  The mode is per process, hence
    subprocesses.  (The bf16x3 / bf16x6 plane modes of rounds 1-2 were removed: slower than f16x3 at equal or lower accuracy.)"""
    import os
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "split_mode_probe.py")
    err = {}
    for mode in ("fp32", "f16x3"):
        env = dict(os.environ, DFVO_CONV_PRECISION=mode)
        out = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        err[mode] = float([l for l in out.stdout.splitlines() if l.startswith("relerr")][-1].split()[1])
    print("   window conv 132 -> 64, max relative error vs torch fp32:", err)
    assert err["fp32"] <= 2e-6       # exact fp32 products, summation order only
    assert err["f16x3"] <= 4e-6      # 22-bit operands, dropped term ~2^-22 per product: the same class


F16_CASES = [
    # name, N, H, W, c0, c1, cout, pad_mode, act, up0
    ("128_128_leaky", 2, 96, 312, 128, 0, 128, 0, 1, 0),
    ("subpixel_64+66_128", 2, 96, 312, 64, 66, 128, 0, 1, 0),
    ("regular_3+128_128", 2, 96, 312, 3, 128, 128, 0, 1, 0),
    ("matching_49_128", 2, 96, 312, 49, 0, 128, 0, 1, 0),
    ("128_64", 2, 96, 312, 128, 0, 64, 0, 1, 0),
    ("64_32", 2, 96, 312, 64, 0, 32, 0, 1, 0),
    ("32_32_odd_size", 2, 97, 301, 32, 0, 32, 0, 1, 0),
    ("decoder_up32+64_32_reflect_elu", 1, 96, 320, 32, 64, 32, 1, 3, 1),
    ("decoder_16_16_reflect_elu", 1, 192, 640, 16, 0, 16, 1, 3, 0),
    ("L4_128_128", 2, 48, 156, 128, 0, 128, 0, 1, 0),
    ("L4_feat_96_96", 2, 48, 156, 96, 0, 96, 0, 1, 0),
    ("L5_subpixel_128+130_128", 2, 24, 78, 128, 130, 128, 0, 1, 0),
    ("L6_subpixel_192+194_128", 2, 12, 39, 192, 194, 128, 0, 1, 0),
    ("L6_regular_3+192_128", 2, 12, 39, 3, 192, 128, 0, 1, 0),
    ("resnet_64_64_relu", 1, 48, 160, 64, 0, 64, 0, 2, 0),
    ("resnet_256_256_relu", 1, 12, 40, 256, 0, 256, 0, 2, 0),
]


@pytest.mark.parametrize("case", F16_CASES, ids=[c[0] for c in F16_CASES])
def test_f16x3_window_conv(gpu, case):
    """DFVO_CONV_PRECISION=f16x3 (conv_win_f16s_kernel): 3x3 window layers with operands split into two f16 planes
    (22 mantissa bits) and three exact products per term, fp32 accumulate -- fp32-class accuracy: the SAME bound as the
    exact fp32 kernel, 4e-6 of max|ref| (in these cases the split kernel's error is the smaller of the two)."""
    name, n, h, w, c0, c1, cout, pad_mode, act, up0 = case
    lib = gpu.lib()
    g = torch.Generator().manual_seed(len(name))
    x0 = torch.randn(n, c0, h, w, generator=g)
    x1 = torch.randn(n, c1, h * (2 if up0 else 1), w * (2 if up0 else 1), generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up0 else x0
    if x1 is not None:
        xin = torch.cat([xin, x1], 1)
    xp = F.pad(xin, (1, 1, 1, 1), mode="reflect" if pad_mode else "constant")
    ref = ref_act(F.conv2d(xp.double(), wt.double(), b.double()), act, 0.1 if act == 1 else 1.0).float()
    errs = {}
    for mode in (b"fp32", b"f16x3"):
        gpu.check(lib.dfvo_set_conv_precision(mode))
        try:
            got = run_conv(gpu, x0, wt, b, 1, (1, 1), pad_mode, act, 0.1 if act == 1 else 1.0, x1=x1, up0=up0)
        finally:
            gpu.check(lib.dfvo_set_conv_precision(b"fp32"))
        err, scale = report("%s %s" % (name, mode.decode()), got, ref)
        errs[mode] = err / scale
    assert errs[b"fp32"] <= 4e-6   # K up to 3500 terms, split-K summation order
    assert errs[b"f16x3"] <= 4e-6


F16_MODE_CASES = [c for c in CASES if c[6] > 2]  # (the one- / two-channel heads stay exact fp32 in every mode)


@pytest.mark.parametrize("case", F16_MODE_CASES, ids=[c[0] for c in F16_MODE_CASES])
def test_conv_f16_mode(gpu, case):
    """DFVO_CONV_PRECISION=f16 (BASELINE config 5's "fp16 flow"): every MFMA kernel family with the cross products compiled
    out.  What the mode computes is DEFINED: both operands rounded to nearest f16, exact products, fp32 accumulation,
    fp32 bias / residual / activation -- so it is checked against the float64 convolution of the ROUNDED operands at the
    fp32 kernels' own tolerance (2e-5: summation order only), and its distance to the unrounded convolution is printed
    (expected ~ 2^-11 relative per operand)."""
    name, n, h, w, c0, c1, cout, kh, kw, stride, pad, pad_mode, act, up0, use_res = case
    lib = gpu.lib()
    g = torch.Generator().manual_seed(hash(name) % 10000)
    x0 = torch.randn(n, c0, h, w, generator=g)
    H, W = (2 * h, 2 * w) if up0 else (h, w)
    x1 = torch.randn(n, c1, H, W, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, kh, kw, generator=g) / np.sqrt((c0 + c1) * kh * kw)
    b = torch.randn(cout, generator=g) * 0.1
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up0 else x0
    if x1 is not None:
        xin = torch.cat([xin, x1], 1)

    def conv64(xx, ww):
        if pad_mode == 1:
            return F.conv2d(F.pad(xx, (pad[1], pad[1], pad[0], pad[0]), mode="reflect"), ww, b.double(), stride=stride)
        return F.conv2d(xx, ww, b.double(), stride=stride, padding=pad)

    res = torch.randn(conv64(xin.double(), wt.double()).shape, generator=g) if use_res else None
    a = 0.1 if act == 1 else 1.0
    ref_rounded = conv64(xin.half().double(), wt.half().double())
    ref_exact = conv64(xin.double(), wt.double())
    if res is not None:
        ref_rounded, ref_exact = ref_rounded + res.double(), ref_exact + res.double()
    ref_rounded, ref_exact = ref_act(ref_rounded, act, a), ref_act(ref_exact, act, a)
    gpu.check(lib.dfvo_set_conv_precision(b"f16"))
    gpu.f16s_overflow_count(reset=True)
    try:
        got = run_conv(gpu, x0, wt, b, stride, pad, pad_mode, act, a, x1, up0, res)
    finally:
        gpu.check(lib.dfvo_set_conv_precision(b"fp32"))
    assert gpu.f16s_overflow_count(reset=True) == 0
    err, scale = report("conv f16 " + name + " vs rounded operands", got, ref_rounded)
    report("conv f16 " + name + " vs the exact convolution", got, ref_exact)
    assert err <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("case", ["tiny_x", "tiny_w", "tiny_both", "mixed_12_decades"])
@pytest.mark.parametrize("layer", ["win_128_128", "gen_L6_192+52_128", "taps_7x1_32_49"])
def test_f16x3_dynamic_range(gpu, case, layer):
    """The f16x3 split outside O(1) data (round-4 verdict).  x = hi + 2^-11 lo with hi = f16(x): for |x| < 2^-14 the hi
    plane is subnormal (absolute spacing 2^-24) and the scaled residue lo = f16((x - hi) 2^11) has absolute spacing 2^-24
    as well, so the split represents such an x to 2^-36 ABSOLUTE, not to 2^-22 relative: an activation of 1e-6 keeps ~16
    bits, below 2^-25 the hi plane is zero and lo alone carries the value (11 bits).  Stated bound, per output:
        |err| <= 2^-22 sum|w x|                      (normal range: 22-bit operands, the dropped lo x lo product)
               + 2^-36 (sum|w| [some |x| < 2^-14] + sum|x| [some |w| < 2^-14])     (subnormal planes: absolute per element)
               + 2^-20 sum|w x|                      (fp32 accumulation of up to a few thousand terms)
    Checked against float64 on a window layer, a K-sliced small-map layer and a tap-window layer; the exact fp32 kernels'
    error on the same data is printed beside it (inputs scaled by 1e-6: fp32 stays at ~2e-6 of max|ref| where f16x3 falls to
    ~1e-5 -- the one regime where the split is not fp32-class; no layer of either net runs there: activations are O(0.1-10),
    asserted indirectly by the nets' parity tests and directly by the range counter for the upper end)."""
    lib = gpu.lib()
    n, h, w, c0, c1, cout, kh, kw, pad = {"win_128_128": (2, 96, 160, 128, 0, 128, 3, 3, (1, 1)),
                                          "gen_L6_192+52_128": (2, 6, 19, 192, 52, 128, 3, 3, (1, 1)),
                                          "taps_7x1_32_49": (2, 96, 160, 32, 0, 49, 7, 1, (3, 0))}[layer]
    g = torch.Generator().manual_seed(len(case) * 7 + len(layer))
    x = torch.randn(n, c0 + c1, h, w, generator=g)
    wt = torch.randn(cout, c0 + c1, kh, kw, generator=g) / np.sqrt((c0 + c1) * kh * kw)
    if case in ("tiny_x", "tiny_both"):
        x = x * 1e-6
    if case in ("tiny_w", "tiny_both"):
        wt = wt * 1e-3
    if case == "mixed_12_decades":  # magnitudes log-uniform over 1e-8 .. 1e4 in both operands
        x = x * torch.pow(10.0, torch.rand(x.shape, generator=g) * 12 - 8)
        wt = wt * torch.pow(10.0, torch.rand(wt.shape, generator=g) * 6 - 5)
    x0, x1 = x[:, :c0].contiguous(), (x[:, c0:].contiguous() if c1 else None)
    b = torch.zeros(cout)
    ref = F.conv2d(x.double(), wt.double(), padding=pad)
    sabs = F.conv2d(x.abs().double(), wt.abs().double(), padding=pad)            # sum |w x| per output
    sw = F.conv2d(torch.ones_like(x).double(), wt.abs().double(), padding=pad)   # sum |w| over the taps inside the image
    sx = F.conv2d(x.abs().double(), torch.ones_like(wt).double(), padding=pad)   # sum |x|
    xt, wtiny = float(x.abs().min()) < 2.0 ** -14, float(wt.abs().min()) < 2.0 ** -14
    bound = 2.0 ** -22 * sabs + 2.0 ** -20 * sabs + 2.0 ** -36 * ((sw if xt else 0) + (sx if wtiny else 0))
    out = {}
    for mode in (b"fp32", b"f16x3"):
        gpu.check(lib.dfvo_set_conv_precision(mode))
        gpu.f16s_overflow_count(reset=True)
        try:
            out[mode] = run_conv(gpu, x0, wt, b, 1, pad, 0, 0, 0.0, x1=x1).double()
        finally:
            gpu.check(lib.dfvo_set_conv_precision(b"fp32"))
        assert gpu.f16s_overflow_count(reset=True) == 0
    e32, e16 = (out[b"fp32"] - ref).abs(), (out[b"f16x3"] - ref).abs()
    scale = float(ref.abs().max())
    print("   f16x3 range %-18s %-16s max|ref| %.2e  max err: fp32 %.2e (rel %.1e)  f16x3 %.2e (rel %.1e)  worst err / bound %.3f"
          % (layer, case, scale, float(e32.max()), float(e32.max()) / scale, float(e16.max()), float(e16.max()) / scale,
             float((e16 / bound).max())))
    assert bool((e16 <= bound).all())


@pytest.mark.gpu
def test_conv_small_maps_with_k_divided_over_workgroups(gpu):
    """DFVO_F16G_NZ=-1 (off by default: slower inside the pipeline): the small-map cases again with the K range divided over
    workgroups and handed over through the split-K workspace -- the environment switch is read once per process, hence a
    child pytest over exactly those cases"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DFVO_F16G_NZ="-1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "test_conv and not f16_mode and (resnet_512_small or resnet_256_12x40 or flow_L5 or flow_L6 or 192_out or dec_refl_up_cat_elu)"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:]
    assert "14 passed" in r.stdout, r.stdout[-500:]  # seven cases x two precisions


# ---- round 6: window launches with tiles of two heights are bit-identical to single-height launches ---------------------
_MIX_WORKER = r'''
import importlib, os, sys, zlib
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
importlib.import_module("df-vo_amd")
capi = importlib.import_module("df-vo_amd.capi")
from test_ops_gpu import run_conv
capi.check(capi.lib().dfvo_set_conv_precision(b"f16x3"))
g = torch.Generator().manual_seed(11)
# shapes whose single-height launch leaves a mostly empty last round (so that the mixed rule fires), odd heights, 1-3 samples,
# one- and two-source layers, 64 and 128 couts, reflection padding + nearest-x2 upsampled first source (the depth decoder's form)
for name, n, h, w, c0, c1, cout, pad_mode, up0 in [("a", 2, 176, 608, 128, 0, 64, 0, 0), ("b", 2, 176, 608, 64, 66, 128, 0, 0), ("c", 1, 203, 640, 64, 0, 64, 0, 0),
                                                 ("d", 3, 131, 333, 32, 16, 128, 0, 0), ("e", 1, 96, 320, 64, 64, 64, 1, 1), ("f", 2, 181, 500, 49, 0, 128, 0, 0)]:
    hs, ws = (h // 2, w // 2) if up0 else (h, w)
    x0 = torch.randn(n, c0, hs, ws, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.05
    out = run_conv(capi, x0, wt, torch.randn(cout, generator=g), 1, (1, 1), pad_mode, 1, 0.1, x1=x1, up0=up0)
    print("CRC", name, "%%08x" %% (zlib.crc32(out.numpy().tobytes()) & 0xffffffff), flush=True)
assert capi.f16s_overflow_count() == 0
'''


def test_mixed_height_window_launches_are_bit_identical(gpu, tmp_path):
    """conv_win_f16s2_mix_kernel (csrc/conv_win_f16s2.h, round 6): tall tiles for the first rows of every sample, short ones for
    the rest, in ONE launch.  Every output pixel runs the same instruction sequence whatever tile it falls into, so the layer's
    output must equal the single-height launch's bit for bit: the same six layers with DFVO_WIN_MIX=1 (default) and =0, and on
    three-row / two-row tiles only (DFVO_WIN_FORCE_TR), in separate processes (the switches are read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "mix_worker.py"
    script.write_text(_MIX_WORKER % {"root": root})
    crcs = {}
    for tag, env in (("mix", {"DFVO_WIN_MIX": "1"}), ("single", {"DFVO_WIN_MIX": "0"}), ("tr3", {"DFVO_WIN_FORCE_TR": "3"}), ("tr2", {"DFVO_WIN_FORCE_TR": "2"})):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        crcs[tag] = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("CRC")]
        assert len(crcs[tag]) == 6, r.stdout
    print("   ", crcs["mix"])
    assert crcs["mix"] == crcs["single"] == crcs["tr3"] == crcs["tr2"]
