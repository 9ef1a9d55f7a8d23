"""Helper of test_split_precision_modes (run in a subprocess: the conv precision mode is read once per process from
DFVO_CONV_PRECISION).  One LDS-window layer (two sources, 132 -> 64 channels, ragged tile) against torch fp32; prints
`relerr <max |err| / max |ref|>`."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import importlib  # noqa: E402

capi = importlib.import_module("df-vo_amd.capi")
from test_ops_gpu import run_conv  # noqa: E402

g = torch.Generator().manual_seed(7)
x0 = torch.randn(2, 4, 99, 157, generator=g)
x1 = torch.randn(2, 128, 99, 157, generator=g)
wt = torch.randn(64, 132, 3, 3, generator=g) / np.sqrt(132 * 9)
b = torch.randn(64, generator=g) * 0.1
ref = F.leaky_relu(F.conv2d(torch.cat([x0, x1], 1), wt, b, padding=1), 0.1)
got = run_conv(capi, x0, wt, b, 1, (1, 1), 0, 1, 0.1, x1, 0, None)
print("relerr %.6e" % (float((got - ref).abs().max()) / float(ref.abs().max())))
