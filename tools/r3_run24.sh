#!/bin/bash
# round 3, GPU call 24: A/B of the two copy changes (same box, alternating)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
one() {
  DFVO_FLOW_DIRECT_OUT=$1 DFVO_CARRY_ONE_LAUNCH=$2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct $1 one_launch $2:', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], '| recomputed', d['features_recomputed']['value'])"
}
for rep in 1 2 3; do one 0 0; one 0 1; one 1 0; one 1 1; done | tee gpurun_out/r3x_copy_ab.txt
