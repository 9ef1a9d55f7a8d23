#!/bin/bash
# round 3, GPU call 15: track() split into begin / end (the host feeds the nets while the chain runs): parity + rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_trajectory_gpu.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r3o_tests.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('begin/end', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
done | tee -a gpurun_out/r3o_tests.txt
DFVO_TRACK_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>&1 >/dev/null | grep -E "track" | tail -4
