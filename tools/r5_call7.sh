#!/bin/bash
# round 5: single-layer counters of the level-2 128 -> 128 window layer in f16x3 (three products) and f16 (one), same tiles
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r5f_pmc_window_f16x3_vs_f16.txt; : > $O
for prec in f16x3 f16; do
  echo "== DFVO_CONV_PRECISION=$prec, unprofiled" >> $O
  DFVO_CONV_PRECISION=$prec N=2 H=176 W=608 C0=128 COUT=128 K=3 ITERS=20 timeout 120 python tools/bench_conv.py 2>/dev/null | grep "^conv" >> $O
  echo "== DFVO_CONV_PRECISION=$prec, rocprofv3 --pmc passes" >> $O
  DFVO_CONV_PRECISION=$prec LAYER="N=2 H=176 W=608 C0=128 COUT=128 K=3" bash tools/pmc_layer.sh "L2 128->128 $prec" 2>/dev/null >> $O
done
cat $O | cut -c1-600
