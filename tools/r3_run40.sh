#!/bin/bash
# round 3, GPU call 40: headline vs the number of untimed warm-up pairs (the in-process later legs run ~1.5 % faster than the first)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for w in 5 20 60; do
  timeout 300 python bench.py --steps 30 --warmup $w --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $w', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'third leg', d['features_recomputed'] and d['features_recomputed']['value'])"
done; done | tee gpurun_out/r3al_warmup.txt
