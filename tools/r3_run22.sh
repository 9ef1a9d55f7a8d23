#!/bin/bash
# round 3, GPU call 22: carry copies in one launch, the flow net writes the slot buffers itself (no copies behind a pass): parity + rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_e2e_gpu.py tests/test_trajectory_gpu.py tests/test_dropin_gpu.py tests/test_nets_gpu.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r3v_tests.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('carried', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], '| recomputed', d['features_recomputed']['value'])"
done | tee -a gpurun_out/r3v_tests.txt
