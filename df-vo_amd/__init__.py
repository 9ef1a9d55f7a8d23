"""df-vo_amd: MI355X-native (gfx950) implementation of DF-VO's per-frame tracking hot path.

The directory name contains a hyphen (mandated layout), so import it with
``importlib.import_module("df-vo_amd")`` (see ``dfvo_amd()`` in the repo-root ``__graft_entry__``).

Sub-modules
    capi      ctypes binding of lib/libdfvo_hip.so (include/dfvo_hip.h)
    libs.*    mirror of the reference's ``libs.deep_models`` / ``libs.matching`` / ``libs.tracker``
              class surface on top of the C ABI
    overlay   installs the mirror classes under the reference's module names so that the
              reference's ``apis/run.py`` / ``libs/dfvo.py`` run unchanged
    dist      frame-batch data parallel driver (torch.distributed / RCCL)
"""
import os

# The fused pipeline keeps eight HIP streams busy (two flow-net instances, depth net, solver chain and its two side
# streams, two prefetch streams).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue
# serialise, which costs ~20 % of the pair rate.  Must be in the environment before the HIP runtime initialises, so
# it is set (without overriding the user's choice) when the package is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libdfvo_hip.so")
