import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def capi():
    import __graft_entry__ as g
    g.dfvo_amd()
    import importlib
    return importlib.import_module("df-vo_amd.capi")


@pytest.fixture(scope="session")
def gpu(capi):
    """loaded library + a visible device; GPU tests FAIL (not skip) when the HIP path is unavailable"""
    import torch
    capi.require_gpu()
    assert torch.cuda.is_available(), "torch sees no GPU"
    # the oracle's scale recovery runs the INSTALLED scikit-learn; the library defaults to the reference's pin (0.20.3),
    # which differs in one degenerate rule (tests/test_tracker_gpu.py::test_scale_recovery_sklearn_versions covers both)
    import sklearn
    capi.set_sklearn_compat(sklearn.__version__)
    return capi


def _with_precision(gpu, name):
    gpu.check(gpu.lib().dfvo_set_conv_precision(name.encode()))
    gpu.f16s_overflow_count(reset=True)
    yield name
    gpu.check(gpu.lib().dfvo_set_conv_precision(b"fp32"))
    # the f16 planes saturate at +-65504 (conv_win_f16s.h): never silently -- no activation or weight of any net in
    # the suite may have been clamped
    clamped = gpu.f16s_overflow_count(reset=True)
    assert clamped == 0, "%s: %d activations / weights were clamped to +-65504" % (name, clamped)


@pytest.fixture(params=["fp32", "f16x3"])
def conv_precision(request, gpu):
    """arithmetic of the conv layers for every net packed inside the test (dfvo_set_conv_precision): the exact fp32
    MFMA kernels, and the f16x3 split kernels that bench.py's headline uses -- same tolerances for both"""
    yield from _with_precision(gpu, request.param)


@pytest.fixture(params=["fp32", "f16x3", "f16"])
def conv_precision_any(request, gpu):
    """the two fp32-class modes plus the single-product "f16" mode (BASELINE config 5's "fp16 flow"): for tests whose
    assertions hold in ANY net arithmetic -- the solver stage against the oracle chain on the device's own arrays"""
    yield from _with_precision(gpu, request.param)


@pytest.fixture
def f16_mode(gpu):
    """DFVO_CONV_PRECISION=f16 for the nets packed inside the test (tests/test_f16_mode_gpu.py)"""
    yield from _with_precision(gpu, "f16")
