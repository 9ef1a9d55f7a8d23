#!/bin/bash
# round 3, GPU call 11: stream layouts A/B, nets-ahead depth, then the stage trace under the new placement
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for l in 0 1 2 3 0 1 2 3; do
  DFVO_STREAM_LAYOUT=$l timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layout=$l', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done | tee gpurun_out/r3k_layouts.txt
DFVO_TRACK_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2> gpurun_out/r3k_track_trace.txt > /dev/null
grep -E "track (device|host) ms" gpurun_out/r3k_track_trace.txt | tail -6
