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


def test_comm_entry_points_check_their_arguments():
    """the data-parallel exchange of the C ABI (dfvo_comm_*, dfvo_allgather_poses): argument errors are reported through the
    library's error channel before RCCL is touched (no GPU, no communicator needed)"""
    import ctypes as C
    import importlib
    import numpy as np
    capi = importlib.import_module("df-vo_amd.capi")
    lib = capi.lib()
    h = C.c_void_p()
    idb = (C.c_uint8 * 128)()
    assert lib.dfvo_comm_create(idb, 2, 2, C.byref(h)) == -2 and b"dfvo_comm_create" in lib.dfvo_last_error()  # rank >= world
    assert lib.dfvo_comm_create(None, 1, 0, C.byref(h)) == -2
    assert lib.dfvo_comm_unique_id(None) == -2
    rows, out = np.zeros((3, 17)), np.zeros((3, 17))
    cnt = (C.c_int * 1)(3)
    assert lib.dfvo_allgather_poses(None, capi.as_ptr(rows), 3, cnt, capi.as_ptr(out)) == -2  # null communicator
    assert lib.dfvo_allgather_poses_device(None, None, 0, None, None) == -2
    assert lib.dfvo_comm_destroy(None) == 0  # destroying nothing is not an error
    ms = np.zeros(8)
    assert lib.dfvo_tracker_stage_ms(None, capi.as_ptr(ms)) == -2


def test_missing_rccl_is_an_error_not_a_crash():
    """include/dfvo_hip.h: RCCL is bound with dlopen at first use and "its absence is an error, there is no fallback" -- an error
    CODE with a message, that is (round-4 advisor: the message was built from a second dlerror() call, which returns NULL)."""
    import subprocess
    import sys
    code = ("import ctypes, importlib, __graft_entry__ as g\n"
            "g.dfvo_amd(); capi = importlib.import_module('df-vo_amd.capi'); lib = capi.lib()\n"
            "buf = (ctypes.c_uint8 * 128)()\n"
            "rc = lib.dfvo_comm_unique_id(buf)\n"
            "print('RC', rc, lib.dfvo_last_error().decode())\n")
    env = dict(os.environ, DFVO_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RC")][-1]
    assert int(line.split()[1]) < 0 and "dlopen(/nonexistent/librccl.so.1)" in line, line
