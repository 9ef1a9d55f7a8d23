#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6h_elementwise_ab.txt; : > $O
DFVO_REG_HEAD_LDS=0 timeout 200 python tools/crc_flow.py 2>/dev/null | tail -1 >> $O
DFVO_REG_HEAD_LDS=1 timeout 200 python tools/crc_flow.py 2>/dev/null | tail -1 >> $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
run() { lbl=$1; shift
  for i in 1 2; do
  env "$@" timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lbl mirrors', d['value'], d['stage_ms_per_pair'])" >> $O
  done
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-exact-leg --no-other-legs 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('$lbl fused', d['value'], 'steady', d['steady_state']['value'], 'frac', r['frac'])" >> $O
  env "$@" DFVO_SESSION_TRACE=1 timeout 300 python bench.py --surface mirrors --steps 10 --warmup 5 2>&1 | grep "session trace" | tail -2 >> $O
}
run lds0 DFVO_REG_HEAD_LDS=0
run lds1 DFVO_REG_HEAD_LDS=1
cat $O
