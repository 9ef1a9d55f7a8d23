#!/bin/bash
# round 3, GPU call 12: two more stream layouts, the device trajectory composition, the from-images accounting lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for l in 0 4 5 0 4 5; do
  DFVO_STREAM_LAYOUT=$l timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layout=$l', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done | tee gpurun_out/r3l_layouts45.txt
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_trajectory_gpu.py -q -m gpu -x -s 2>&1 | grep -v amdgpu.ids | grep -E "FROM-IMAGES|passed|failed|Error|HIP  |oracle " | tee gpurun_out/r3l_tests.txt
