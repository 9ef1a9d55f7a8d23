#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6p_probe_stability.txt; : > $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
DFVO_STREAM_PROBE_VERBOSE=1 timeout 300 python bench.py --surface mirrors --steps 20 --warmup 5 2>gpurun_out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mirrors run $i', d['value'], d['stage_ms_per_pair']['depth_cnn'], d['stage_ms_per_pair']['deep_inference'])" >> $O
grep "stream pool" gpurun_out/err.txt >> $O
done
for i in 1 2 3 4; do
DFVO_STREAM_PROBE_VERBOSE=1 timeout 300 python bench.py --conv-precision fp32 --steps 20 --warmup 5 --no-cpu-baseline --no-other-legs --no-exact-leg --no-roofline 2>gpurun_out/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fp32 fused run $i', d['value'])" >> $O
grep "stream pool" gpurun_out/err.txt >> $O
done
cat $O
