"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dfvo_hip.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dfvo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfvo_[a-z0-9_]+)\s*\(", src)))


def test_build_and_exports(capi):
    import __graft_entry__ as g
    g.build()
    lib = capi.lib()
    names = declared_symbols()
    assert len(names) > 20
    for n in names:
        assert hasattr(lib, n), "libdfvo_hip.so does not export " + n
        assert n in capi.SIGNATURES, "capi.SIGNATURES lacks " + n
    for n in capi.SIGNATURES:
        assert n in names, n + " bound in capi.py but not declared in include/dfvo_hip.h"


def test_no_gpu_is_loud(capi):
    """the product path must fail loudly without a GPU (no CPU fallback)"""
    lib = capi.lib()
    if lib.dfvo_device_count() > 0:
        return
    import pytest
    with pytest.raises(capi.DfvoError):
        capi.require_gpu()


def test_bf16_plane_split_of_the_opt_in_conv_modes():
    """host part of DFVO_CONV_PRECISION=bf16x3 / bf16x6 (no GPU needed): plane 0 is torch's round-to-nearest-even bfloat16,
    the planes add up to the value to 2^-16 (two planes) / exactly-or-2^-24 (three planes)"""
    import importlib
    import numpy as np
    import torch
    capi = importlib.import_module("df-vo_amd.capi")
    rng = np.random.default_rng(5)
    x = np.ascontiguousarray((rng.standard_normal(4096) * 10.0 ** rng.integers(-4, 4, 4096)).astype(np.float32))
    x[:4] = [0.0, -0.0, 1.0, -3.0e-39]  # zeros and a denormal
    for planes, tol in ((2, 2.0 ** -16), (3, 2.0 ** -23)):
        out = np.zeros(planes * x.size, np.uint16)
        capi.check(capi.lib().dfvo_split_bf16_planes(capi.as_ptr(x), x.size, planes, capi.as_ptr(out)))
        pl = (out.reshape(planes, -1).astype(np.uint32) << 16).view(np.float32)
        want0 = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
        assert np.array_equal(pl[0].view(np.uint32), want0.view(np.uint32))
        rec = pl.astype(np.float64).sum(0)
        assert np.all(np.abs(rec - x.astype(np.float64)) <= tol * np.abs(x.astype(np.float64)) + 1e-37)  # (bf16 denormals are coarse)
