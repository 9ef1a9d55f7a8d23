#!/bin/bash
# round 3, GPU call 16-17: device-gated PnP draws, chains of consecutive pairs enqueued back to back: parity + rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_trajectory_gpu.py tests/test_dropin_gpu.py tests/test_pnp_gpu.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r3q_tests.txt
for ca in 1 0 1 0; do
  DFVO_BENCH_CHAIN_AHEAD=$ca timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain_ahead $ca', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
done | tee -a gpurun_out/r3q_tests.txt
DFVO_TRACK_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>&1 >/dev/null | grep -E "track" | tail -4 | tee -a gpurun_out/r3q_tests.txt
