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
