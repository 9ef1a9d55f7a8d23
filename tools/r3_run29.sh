#!/bin/bash
# round 3, GPU call 29: vectorised regularisation head: bit-identity A/B (and the correlation kernel's), nets tests, kernel timing from a short trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_nets_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in 0 1; do
  cd /tmp; rm -rf /tmp/p_rh$v
  DFVO_REG_HEAD_V=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_rh$v -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-exact-leg --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find /tmp/p_rh$v -name "*kernel_stats.csv" | head -1)
  echo "== DFVO_REG_HEAD_V=$v"; grep -E "k_reg_head|k_correlation|k_deconv_dw|k_warp|k_flow_mean|k_flow_resize|k_flow_consistency|k_reg_prep|k_copy_segments|k_img_u8|k_resize" $f | cut -d, -f1-4 | cut -c1-150
  cd $GRAFT_REPO_ROOT
done
} | tee gpurun_out/r3ab_reg_head.txt
