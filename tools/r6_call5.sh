#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6f_side_graphs_ab.txt; : > $O
DFVO_FLOW_SIDE_BRANCH=0 timeout 200 python tools/crc_flow.py 2>/dev/null | tail -1 >> $O
DFVO_FLOW_SIDE_BRANCH=1 timeout 200 python tools/crc_flow.py 2>/dev/null | tail -1 >> $O
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
run side0 DFVO_FLOW_SIDE_BRANCH=0
run side1 DFVO_FLOW_SIDE_BRANCH=1
cat $O
