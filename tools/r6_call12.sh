#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6n_kp_stage.txt; : > $O
timeout 1500 python -m pytest tests/test_tracker_gpu.py tests/test_dropin_gpu.py tests/test_pipeline_gpu.py tests/test_e2e_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
for i in 1 2; do
timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mirrors', d['value'], d['stage_ms_per_pair'])" >> $O
done
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-exact-leg --no-other-legs 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('fused', d['value'], 'steady', d['steady_state']['value'], 'frac', r['frac'])" >> $O
DFVO_SESSION_TRACE=1 timeout 300 python bench.py --surface mirrors --steps 10 --warmup 5 2>&1 | grep "session trace" | tail -3 >> $O
cat $O
