#!/bin/bash
# round 3, GPU call 31: the fused five-point stages (DFVO_E_FUSED 0 / 1 / 2) re-measured on the track_begin / track_end build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do for f in 0 2 1; do
  DFVO_E_FUSED=$f timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E_FUSED $f', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
done; done | tee gpurun_out/r3ad_e_fused.txt
