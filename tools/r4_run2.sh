#!/bin/bash
# round 4, GPU call 2: full -m gpu suite; rocprofv3 kernel statistics of the default bench with the lane-parallel
# polynomial stage (1) and round 3's (0)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4b_tests.txt
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-roofline --no-exact-leg --steps 40 --warmup 5"
for v in 1 0; do
  DFVO_E_POLY_LANES=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats$v -o r -- $CMD > /tmp/b_stats$v.log 2>&1
  f=$(find /tmp/p_stats$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/r4b_kernel_stats_poly$v.csv
  tail -1 /tmp/b_stats$v.log | cut -c1-200
done
cd $R
tail -6 gpurun_out/r4b_tests.txt
grep -h "k_e_\|k_h_\|k_scale\|k_rep\|k_recover\|k_gric\|k_mt" gpurun_out/r4b_kernel_stats_poly1.csv | cut -c1-150 | head -40
echo ----
grep -h "k_e_" gpurun_out/r4b_kernel_stats_poly0.csv | cut -c1-150
