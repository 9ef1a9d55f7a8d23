#!/bin/bash
# round 3, GPU call 39: headline leg with the pyramids carried vs recomputed, separate processes, alternating (order effects of the in-process legs excluded)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3 4; do for m in on off; do
  timeout 300 python bench.py --feature-carry $m --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('carry $m', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'third leg', d['features_recomputed'] and d['features_recomputed']['value'])"
done; done | tee gpurun_out/r3ak_carry_order.txt
