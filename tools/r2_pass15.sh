#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_e2e_gpu.py tests/test_trajectory_gpu.py tests/test_dropin_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/pass17.log
for cfg in "20 5" "20 5"; do set -- $cfg; echo -n "steps $1 warmup $2: " >> gpurun_out/pass17.log; timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps $1 --warmup $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(d['value'], d['ms_per_step'], 'PnP', c['tracked_by_PnP'])" >> gpurun_out/pass17.log; done
cat gpurun_out/pass17.log
