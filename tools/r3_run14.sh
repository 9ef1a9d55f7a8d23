#!/bin/bash
# round 3, GPU call 14: nets alone vs solver stage alone under the round-3 build (f16x3), fused five-point stages A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
DFVO_CONV_PRECISION=f16x3 STEPS=40 timeout 300 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r3n_stages.txt
for ef in 0 2 0 2; do
  DFVO_E_FUSED=$ef timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E_FUSED=$ef', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
done | tee -a gpurun_out/r3n_stages.txt
