#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
DFVO_E_FUSED=2 timeout 900 python -m pytest tests/test_solvers_gpu.py tests/test_tracker_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/efused.log
for f in 2 0 2; do echo -n "DFVO_E_FUSED=$f: " >> gpurun_out/efused.log; DFVO_E_FUSED=$f DFVO_TRACK_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>gpurun_out/ef.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/efused.log; grep "track device" gpurun_out/ef.err | tail -1 >> gpurun_out/efused.log; done
cat gpurun_out/efused.log
