#!/bin/bash
# round 3, GPU call 30: non-conv flow-net kernels, per-launch time from rocprofv3 kernel statistics, vectorised reg head off / on
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1; do
  cd /tmp; rm -rf /tmp/p_rh$v
  DFVO_REG_HEAD_V=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_rh$v -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-exact-leg --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find /tmp/p_rh$v -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r3ac_stats_reghead$v.csv
  cd $GRAFT_REPO_ROOT
done
python - <<'PY' | tee gpurun_out/r3ac_nonconv.txt
import csv
for v in (0,1):
    rows=list(csv.DictReader(open('gpurun_out/r3ac_stats_reghead%d.csv'%v)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print('== DFVO_REG_HEAD_V=%d total kernel ms %.1f'%(v,tot/1e6))
    for r in rows:
        n=r['Name']
        if any(k in n for k in ('k_reg_head','k_correlation','k_deconv_dw','k_warp','k_flow_mean','k_flow_resize','k_flow_consistency','k_reg_prep','k_copy_segments','k_img_u8','k_resize_bilinear','copyBuffer','conv_head')):
            print('  %-60s calls %5s avg %8.1f us total %7.2f ms'%(n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
