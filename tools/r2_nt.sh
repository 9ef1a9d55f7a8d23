#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( echo "== loaded (bench.py default)"; DFVO_TRACK_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 80 --warmup 10 2>&1 >/dev/null | grep "track " | tail -4
  echo "== unloaded (tools/bench_stages.py: solver stage alone, no prefetch)"; DFVO_TRACK_TRACE=1 DFVO_CONV_PRECISION=f16x3 STEPS=40 timeout 300 python tools/bench_stages.py 2>&1 | grep "track \|solver" | tail -4 ) > gpurun_out/chain_segments.log
cat gpurun_out/chain_segments.log
