"""CPU (build container: needs /root/reference): the drop-in boundary as a class surface.  For every class the
overlay substitutes, each public method of the reference's class -- and of the base class it inherits its interface from --
must exist on the mirror under the same name with the same positional parameters.  Training-only methods are listed
explicitly: they must exist nowhere or raise NotImplementedError; nothing is silently absent."""
import ast
import importlib
import inspect
import os

import pytest

REF = "/root/reference"
PAIRS = [  # (reference file, reference classes whose methods form the interface, mirror module, mirror class)
    ("libs/deep_models/deep_models.py", ["DeepModel"], "df-vo_amd.libs.deep_models.deep_models", "DeepModel"),
    ("libs/deep_models/flow/lite_flow_net/lite_flow.py", ["LiteFlow"], "df-vo_amd.libs.deep_models.flow.lite_flow_net.lite_flow", "LiteFlow"),
    ("libs/deep_models/flow/deep_flow.py", ["DeepFlow"], "df-vo_amd.libs.deep_models.flow.lite_flow_net.lite_flow", "LiteFlow"),
    ("libs/deep_models/depth/monodepth2/monodepth2.py", ["Monodepth2DepthNet"], "df-vo_amd.libs.deep_models.depth.monodepth2.monodepth2", "Monodepth2DepthNet"),
    ("libs/deep_models/depth/deep_depth.py", ["DeepDepth"], "df-vo_amd.libs.deep_models.depth.monodepth2.monodepth2", "Monodepth2DepthNet"),
    ("libs/matching/keypoint_sampler.py", ["KeypointSampler"], "df-vo_amd.libs.matching.keypoint_sampler", "KeypointSampler"),
    ("libs/tracker/E_tracker.py", ["EssTracker"], "df-vo_amd.libs.tracker.E_tracker", "EssTracker"),
    ("libs/tracker/pnp_tracker.py", ["PnpTracker"], "df-vo_amd.libs.tracker.pnp_tracker", "PnpTracker"),
    ("libs/geometry/camera_modules.py", ["SE3", "Intrinsics"], "df-vo_amd.libs.geometry.camera_modules", None),
]
# loss / optimiser plumbing of online finetuning (out of scope) -- internal to the reference's own train()
TRAINING_INTERNALS = {"train", "generate_images_pred_flow", "compute_flow_losses", "compute_reprojection_loss", "reprojection",
                      "warp_image", "compute_depth_loss", "compute_depth_consistency_losses"}


def _ref_methods(path, cls):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name == cls:
            out = {}
            for m in n.body:
                if isinstance(m, ast.FunctionDef) and not m.name.startswith("_"):
                    is_prop = any(isinstance(d, ast.Name) and d.id == "property" or isinstance(d, ast.Attribute) and d.attr == "setter"
                                  for d in m.decorator_list)
                    out.setdefault(m.name, ("property" if is_prop else "method", [a.arg for a in m.args.args][1:]))
            return out
    raise AssertionError("class %s not found in %s" % (cls, path))


@pytest.mark.parametrize("ref_file,ref_classes,mod,mirror_cls", PAIRS, ids=[p[0].split("/")[-1] + ":" + "+".join(p[1]) for p in PAIRS])
def test_mirror_has_the_reference_surface(ref_file, ref_classes, mod, mirror_cls):
    if not os.path.isdir(REF):
        pytest.skip("/root/reference is only present in the build container")
    m = importlib.import_module(mod)
    checked = 0
    for rc in ref_classes:
        mirror = getattr(m, mirror_cls or rc)
        for name, (kind, params) in _ref_methods(ref_file, rc).items():
            if name in TRAINING_INTERNALS:
                continue
            assert hasattr(mirror, name), "%s.%s is missing from the mirror" % (rc, name)
            attr = inspect.getattr_static(mirror, name)
            if kind == "property":
                assert isinstance(attr, property) and attr.fset is not None, "%s.%s must be a read / write property" % (rc, name)
            else:
                fn = attr.__func__ if isinstance(attr, (staticmethod, classmethod)) else attr
                got = list(inspect.signature(fn).parameters)[1:]
                assert got[:len(params)] == params, "%s.%s%s: the mirror takes %s" % (rc, name, tuple(params), tuple(got))
            checked += 1
    assert checked >= 2
