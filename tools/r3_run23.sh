#!/bin/bash
# round 3, GPU call 23: rate only, five repetitions (was call 22's spread the box or the build?)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('carried', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], '| recomputed', d['features_recomputed']['value'])"
done | tee gpurun_out/r3w_rate.txt
