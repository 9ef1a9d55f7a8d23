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
    return capi
