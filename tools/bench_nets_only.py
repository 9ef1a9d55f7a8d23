"""Nets only (both CNNs through the fused pipeline object, graphs on): for kernel traces of the net passes alone."""
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
    syn = importlib.import_module("df-vo_amd.synthetic")
    pmod = importlib.import_module("df-vo_amd.pipeline")
    H, W, n = 376, 1241, int(os.environ.get("STEPS", 10))
    K = syn.rigid_scene(64, 64, seed=1)["K"]
    pipe = pmod.TrackingPipeline(H, W, 192, 640, K, syn.liteflownet_state_dict(4869), syn.monodepth2_state_dict(4869))
    ref, cur = syn.image_pair(H, W, seed=1)
    feed = np.asarray(Image.fromarray(cur).resize((640, 192), Image.LANCZOS))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_ref, d_cur, d_feed = dev(ref), dev(cur), dev(feed)
    for k in range(3):
        pipe.enqueue_nets(k % 4, d_ref, d_cur, d_feed)
    pipe.sync()
    if os.environ.get("SAMPLE"):  # shader clock / power while the loop below runs (rocm-smi from a side thread)
        import subprocess
        import threading

        def sample():
            time.sleep(1.5)
            for _ in range(6):
                out = subprocess.run("rocm-smi --showclocks --showpower 2>&1 | grep -i 'sclk\\|Power (W)' | tr '\n' ' '",
                                     shell=True, capture_output=True, text=True).stdout
                print("  smi:", " ".join(out.split()), flush=True)
        threading.Thread(target=sample, daemon=True).start()
    t0 = time.perf_counter()
    for k in range(n):
        pipe.enqueue_nets(k % 4, d_ref, d_cur, d_feed)
    pipe.sync()
    print("nets only: %.3f ms/pair" % ((time.perf_counter() - t0) / n * 1e3))
    pipe.close()


if __name__ == "__main__":
    main()
