#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_dropin_gpu.py tests/test_multigpu_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r6d_dropin_tests.txt
cat gpurun_out/r6d_dropin_tests.txt
for i in 1 2; do
timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>gpurun_out/r6d_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mirrors', d['value'], d['stage_ms_per_pair'], d['session'])"
grep "stream pool" gpurun_out/r6d_err.txt | tail -3
done
DFVO_STREAM_PROBE_VERBOSE=1 timeout 300 python bench.py --surface mirrors --steps 10 --warmup 3 2>&1 | grep "stream pool" | tail -6
