"""CPU: df-vo_amd/overlay.py really wires the reference's UNCHANGED libs/dfvo.py to the mirror classes.

overlay.install() registers the mirrors under the reference's module names; the reference's `libs.dfvo` is then imported
from /root/reference (third-party imports shimmed by oracle/ref_shims, cv2 by the oracle's cv2_shim) and
  * its three hot-path imports (libs/dfvo.py:24,27,28) must resolve to df-vo_amd.libs.*,
  * DFVO.initialize_tracker (libs/dfvo.py:95-107) must construct the mirrors,
  * everything else (Timer, SE3, datasets, FrameDrawer) must still be the reference's own code.
Runs in a subprocess so that the module aliases do not leak into the test session; skipped where /root/reference does
not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import importlib, os, sys, types
ROOT, REF = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shims"))
sys.path.insert(0, REF)
import numpy as np
if not hasattr(np, "int"):
    np.int = int
from oracle import cv2_shim
sys.modules["cv2"] = cv2_shim
import matplotlib
matplotlib.use("Agg")
importlib.import_module("df-vo_amd")
overlay = importlib.import_module("df-vo_amd.overlay")
names = overlay.install()
assert "libs.tracker" in names and "libs.deep_models.deep_models" in names
import libs.dfvo as ref_dfvo                      # the reference's file, unmodified
assert os.path.realpath(ref_dfvo.__file__) == os.path.realpath(os.path.join(REF, "libs", "dfvo.py")), ref_dfvo.__file__
mine = lambda m: importlib.import_module("df-vo_amd." + m)
assert ref_dfvo.DeepModel is mine("libs.deep_models.deep_models").DeepModel
assert ref_dfvo.KeypointSampler is mine("libs.matching.keypoint_sampler").KeypointSampler
assert ref_dfvo.EssTracker is mine("libs.tracker").EssTracker and ref_dfvo.PnpTracker is mine("libs.tracker").PnpTracker
# the rest is still the reference's own code
assert os.path.realpath(sys.modules["libs.general.timer"].__file__).startswith(os.path.realpath(REF))
assert os.path.realpath(sys.modules["libs.geometry.camera_modules"].__file__).startswith(os.path.realpath(REF))
# DFVO.initialize_tracker / the sampler construction of DFVO.__init__ with the mirrors, on the reference's own objects
cfg_mod = mine("default_cfg")
cfg = cfg_mod.default_configuration(376, 1241, "flow.pth", "depth_dir")
vo = ref_dfvo.DFVO.__new__(ref_dfvo.DFVO)
vo.cfg = cfg
vo.timers = ref_dfvo.Timer()
cam = sys.modules["libs.geometry.camera_modules"].Intrinsics([607.19, 185.22, 718.856, 718.856])
vo.dataset = types.SimpleNamespace(cam_intrinsics=cam)
vo.tracking_method = cfg.tracking_method
vo.initialize_tracker()
assert type(vo.e_tracker) is mine("libs.tracker.E_tracker").EssTracker and vo.e_tracker.timers is vo.timers
assert type(vo.pnp_tracker) is mine("libs.tracker.pnp_tracker").PnpTracker
vo.kp_sampler = ref_dfvo.KeypointSampler(vo.cfg)
vo.deep_models = ref_dfvo.DeepModel(vo.cfg)      # (initialize_models() needs the GPU: tests/test_dropin_gpu.py)
assert type(vo.deep_models) is mine("libs.deep_models.deep_models").DeepModel
print("OVERLAY-OK", len(names))
'''


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "libs", "dfvo.py")), reason="reference checkout not present (GPU box)")
def test_overlay_wires_the_unmodified_reference_to_the_mirrors():
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, REF], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    assert r.returncode == 0 and "OVERLAY-OK" in r.stdout, r.stdout[-3000:]
