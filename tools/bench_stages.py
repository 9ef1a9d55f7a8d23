"""Stage timing of the fused pipeline: nets only (throughput) and solver stage only (latency), unloaded."""
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from PIL import Image
    capi = importlib.import_module("df-vo_amd.capi")
    syn = importlib.import_module("df-vo_amd.synthetic")
    pmod = importlib.import_module("df-vo_amd.pipeline")
    H, W, n = 376, 1241, int(os.environ.get("STEPS", 20))
    scenes = [syn.rigid_scene(H, W, seed=100 + i) for i in range(4)]
    pipe = pmod.TrackingPipeline(H, W, 192, 640, scenes[0]["K"], syn.liteflownet_state_dict(4869),
                                 syn.monodepth2_state_dict(4869), seed=4869)
    ref, cur = syn.image_pair(H, W, seed=1)
    feed = np.asarray(Image.fromarray(cur).resize((640, 192), Image.LANCZOS))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_ref, d_cur, d_feed = dev(ref), dev(cur), dev(feed)
    d_sc = [(dev(s["flow"]), dev(s["diff"]), dev(s["depth_cur"])) for s in scenes]
    for k in range(3):
        pipe.enqueue_nets(k % 2, d_ref, d_cur, d_feed)
        pipe.track(k % 2, *d_sc[k % 4])
    pipe.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        pipe.enqueue_nets(k % 2, d_ref, d_cur, d_feed)
    pipe.sync()
    torch.cuda.synchronize()
    t_n = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for k in range(n):
        pipe.track(1 - (n % 2), *d_sc[k % 4])
    pipe.sync()
    t_t = (time.perf_counter() - t0) / n
    print("nets only: %.3f ms/pair   solver stage only: %.3f ms/pair" % (t_n * 1e3, t_t * 1e3))
    pipe.close()


if __name__ == "__main__":
    main()
