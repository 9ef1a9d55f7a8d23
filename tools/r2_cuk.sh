#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/cuk.log
for cfg in "4 7" "8 7" "4 6"; do set -- $cfg; echo -n "DFVO_SOLVER_CU_K=$1 DFVO_SOLVER_CU_ONLY=$2: " >> gpurun_out/cuk.log; DFVO_SOLVER_CU_K=$1 DFVO_SOLVER_CU_ONLY=$2 DFVO_TRACK_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>gpurun_out/cuk.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> gpurun_out/cuk.log; grep "track host" gpurun_out/cuk.err | tail -1 >> gpurun_out/cuk.log; done
cat gpurun_out/cuk.log
