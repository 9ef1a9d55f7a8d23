#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6k_window_presplit_timing.txt; : > $O
for rep in 1 2; do
for v in 0 1; do
echo "== DFVO_WIN_PRESPLIT_TIMING=$v (run $rep)" >> $O
DFVO_WIN_PRESPLIT_TIMING=$v timeout 300 python tools/bench_window_layers.py 2>/dev/null | grep -E "L2 128->128|L2 64\+66|L2 49->128|L3 128->128|c5 128|sum of" >> $O
done
done
cat $O
