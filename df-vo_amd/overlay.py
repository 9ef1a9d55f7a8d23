"""Drop-in overlay: makes the reference's unchanged `apis/run.py` / `libs/dfvo.py` use the MI355X
classes.  `install()` registers this package's mirrors under the reference's module names, so that

    from libs.deep_models.deep_models import DeepModel          (libs/dfvo.py:24)
    from libs.matching.keypoint_sampler import KeypointSampler  (libs/dfvo.py:27)
    from libs.tracker import EssTracker, PnpTracker             (libs/dfvo.py:28)

resolve to df-vo_amd.libs.*, while every other `libs.*` module (dfvo, datasets, general, geometry, ...)
still comes from the reference checkout on sys.path.  See INTEGRATION.md.
"""
import importlib
import sys

PKG = __name__.rsplit(".", 1)[0]

_MAP = {
    "libs.deep_models.deep_models": ".libs.deep_models.deep_models",
    "libs.matching.keypoint_sampler": ".libs.matching.keypoint_sampler",
    "libs.tracker": ".libs.tracker",
    "libs.tracker.E_tracker": ".libs.tracker.E_tracker",
    "libs.tracker.pnp_tracker": ".libs.tracker.pnp_tracker",
}


def install():
    for ref_name, ours in _MAP.items():
        sys.modules[ref_name] = importlib.import_module(ours, PKG)
    return sorted(_MAP)
