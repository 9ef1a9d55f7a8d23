"""CRC of the fused pipeline's net outputs (forward / backward flow, consistency, depth) for one coded KITTI-size pair: the
bit-identity check between kernel variants selected by environment knobs (run once per setting, compare the lines)."""
import importlib
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("df-vo_amd")
capi = importlib.import_module("df-vo_amd.capi")
syn = importlib.import_module("df-vo_amd.synthetic")
pmod = importlib.import_module("df-vo_amd.pipeline")
smod = importlib.import_module("df-vo_amd.sequence")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
capi.check(capi.lib().dfvo_set_conv_precision(prec.encode()))
H, W = 376, 1241
seq = syn.coded_tunnel_sequence(H, W, 2, mode="pot", step=1.0, seed=7, poses=syn.tunnel_poses_lateral(2, 0.4))
pipe = pmod.TrackingPipeline(H, W, 192, 640, seq["K"], syn.crafted_liteflownet_state_dict(H, W, "pot"),
                             syn.crafted_monodepth2_state_dict(), seed=4869)
fr = smod.frames_to_device(seq["frames"])
pipe.set_ref_image(fr[0])
pipe.enqueue_nets(0, fr[0], fr[1])
out = pipe.track(0)
fwd, bwd, diff, raw, dep = pipe.get_outputs(0)
crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff
print("%s crc fwd %08x bwd %08x diff %08x depth %08x | status %d kp %d" % (prec, crc(fwd), crc(bwd), crc(diff), crc(raw), out.status, out.n_kp))
pipe.close()
