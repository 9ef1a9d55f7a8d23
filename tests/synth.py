"""seeded synthetic inputs: re-export of df-vo_amd/synthetic.py for the tests"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_s = importlib.import_module("df-vo_amd.synthetic")
smooth_noise = _s.smooth_noise
image_pair = _s.image_pair
two_view = _s.two_view
rigid_scene = _s.rigid_scene
coded_tunnel_sequence = _s.coded_tunnel_sequence
crafted_liteflownet_state_dict = _s.crafted_liteflownet_state_dict
crafted_monodepth2_state_dict = _s.crafted_monodepth2_state_dict
tunnel_truth = _s.tunnel_truth
tunnel_cast = _s.tunnel_cast
write_weight_files = _s.write_weight_files
tunnel_poses_lateral = _s.tunnel_poses_lateral
