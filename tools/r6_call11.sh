#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6l_small_map_tiles.txt; : > $O
for v in 0 1 2 4; do
echo "== DFVO_F16S2_SMALL=$v" >> $O
DFVO_F16S2_SMALL=$v DFVO_CONV_PROFILE_CSV=/tmp/cp_$v.csv timeout 300 python tools/bench_window_layers.py 2>/dev/null | grep -E "^s |L3 " >> $O
done
cat $O
