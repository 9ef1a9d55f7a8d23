#!/bin/bash
# round 3, GPU call 13: planar PnP initialisation on the device + the refactored homography refinement block
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_pnp_gpu.py tests/test_solvers_gpu.py tests/test_tracker_gpu.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r3m_tests.txt
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
