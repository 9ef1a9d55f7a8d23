#!/bin/bash
# round 3, GPU call 8: does the pair rate still depend on who touched the GPU first?  then the profile passes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for tf in "" 1 "" 1; do
  DFVO_BENCH_TORCH_FIRST=$tf timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torch_first=[$tf]', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done | tee gpurun_out/r3h_torch_first_ab.txt
TAG=r3h bash tools/r3_profile.sh
