#!/bin/bash
# round 4, GPU call 3: third window skeleton (tile runs, next tile's first chunk prefetched in the last chunk): per-layer
# timing + CRCs for run lengths 1 / 2 / 3 / persistent against the second skeleton, then the pair rate for each
mkdir -p gpurun_out
O=gpurun_out/r4c_tile_run_layers.txt; : > $O
for r in "" 1 2 3 p; do
  echo "== DFVO_F16S_RUN='$r'" >> $O
  if [ -z "$r" ]; then timeout 300 python tools/bench_f16s_v2.py >> $O 2>&1; else DFVO_F16S_RUN=$r timeout 300 python tools/bench_f16s_v2.py >> $O 2>&1; fi
done
P=gpurun_out/r4c_tile_run_pairs.txt; : > $P
for r in "" 1 2 3 p "" 2 p; do
  echo "== DFVO_F16S_RUN='$r'" >> $P
  ( if [ -z "$r" ]; then timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg; else DFVO_F16S_RUN=$r timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg; fi ) 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('frames/s', d['value'], 'steady', d['steady_state']['value'], '| dominant', r['kernel'][:40], 'frac', r['frac'], 'avg us', r['avg_launch_us'], 'conv family ms', r['conv_family_ms_per_pair'])
for c in r['by_config'][:3]: print('   ', c['kernel'][:60], c['ms_per_pair'], c['launches_per_pair'], c['tflops'])" >> $P
done
cat $O | grep -v "^$" | cut -c1-110
cat $P
