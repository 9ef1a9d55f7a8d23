#!/bin/bash
# round 3, GPU call 35: every flow pass ordered behind the previous pass's feature stage: carry tests (incl. a full pass right behind a carried one), pipeline / e2e / trajectory, rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_e2e_gpu.py tests/test_trajectory_gpu.py tests/test_dropin_gpu.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rate', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], '| recomputed', d['features_recomputed']['value'])"
done
} | tee gpurun_out/r3ah_order.txt
