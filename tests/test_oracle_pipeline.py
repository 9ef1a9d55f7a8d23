"""CPU: oracle/pipeline_np.py (the image-level frame loop used as the expected value of the end-to-end GPU tests and as
bench.py's CPU baseline) against the fixture written by the REFERENCE's own frame loop -- apis/run.py + DFVO.main /
deep_model_inference / tracking / update_global_pose and its DeepModel / KeypointSampler / EssTracker / PnpTracker
classes, unmodified, weights loaded from files through a stub Dataset (tests/golden/make_golden.py dfvo_main)."""
import os

import numpy as np

from oracle import pipeline_np as P
from synth import coded_tunnel_sequence, crafted_liteflownet_state_dict, crafted_monodepth2_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dfvo_main.npz")


def test_oracle_frame_loop_equals_reference_dfvo_main():
    fx = np.load(GOLD)
    h, w, n = int(fx["h"]), int(fx["w"]), int(fx["n_frames"])
    seq = coded_tunnel_sequence(h, w, n, mode="mux", step=1.0, seed=int(fx["seq_seed"]))
    assert np.array_equal(seq["poses"], fx["gt"])
    r = P.track_sequence(list(seq["frames"]), crafted_liteflownet_state_dict(h, w, "mux"), crafted_monodepth2_state_dict(),
                         seq["K"], seed=4869)
    want_modes = [m for m in fx["modes"][1:]]  # frame 0 has no pair
    got_modes = [{"E": "Ess. Mat.", "PnP": "PnP", "constant_motion": "Ess. Mat."}[s] for s in r["status"]]
    assert got_modes == want_modes
    assert np.array_equal(r["poses"], fx["poses"]), np.abs(r["poses"] - fx["poses"]).max()
    st = np.random.get_state()
    assert np.array_equal(st[1], fx["rng_after"][:624]) and st[2] == int(fx["rng_after"][624])
    # the trajectory file the reference wrote parses back to the same poses (KITTI format, utils.py:329-355)
    lines = str(fx["traj_txt"]).strip().splitlines()
    back = np.array([[float(v) for v in l.split()[1:]] for l in lines]).reshape(n, 3, 4)
    assert np.allclose(back, fx["poses"][:, :3, :], rtol=0, atol=1e-12)
    # and the loop really tracked the rendered camera
    rel = np.linalg.inv(fx["gt"][0]) @ fx["gt"][-1]
    assert np.linalg.norm(fx["poses"][-1][:3, 3] - rel[:3, 3]) < 0.05 * np.linalg.norm(rel[:3, 3])
