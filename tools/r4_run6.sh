#!/bin/bash
mkdir -p gpurun_out
DFVO_CONV_PROFILE_CSV=gpurun_out/r4f_layers.csv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-leg > /dev/null 2>&1
wc -l gpurun_out/r4f_layers.csv; head -3 gpurun_out/r4f_layers.csv
