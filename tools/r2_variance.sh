#!/bin/bash
# run-to-run spread of the default bench line and the exact-fp32 figure of the same build
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/variance.log
for i in 1 2 3; do echo -n "default (f16x3) 20+5 run $i: " >> gpurun_out/variance.log; timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(d['value'], d['ms_per_step'], 'E', c['tracked_by_E'], 'PnP', c['tracked_by_PnP'])" >> gpurun_out/variance.log; done
echo -n "exact fp32 60+10: " >> gpurun_out/variance.log; timeout 300 python bench.py --no-cpu-baseline --conv-precision fp32 --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']; print(d['value'], d['ms_per_step'], 'E', c['tracked_by_E'], 'PnP', c['tracked_by_PnP'], 'dominant', r['kernel'], r['achieved'], r['frac'], 'family', r['conv_family_achieved'], r['conv_family_ms_per_pair'])" >> gpurun_out/variance.log
cat gpurun_out/variance.log
