#!/bin/bash
# round 4: ragged-cout vector epilogue + default tile-run policy: op / net parity (both precisions), then the pair rate
# against round 3's window path (DFVO_F16S_RUN=0), alternating
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r4g_tests.txt
cat gpurun_out/r4g_tests.txt
O=gpurun_out/r4g_policy_ab.txt; : > $O
for r in "" 0 "" 0; do
  echo "== DFVO_F16S_RUN='$r'" >> $O
  ( if [ -z "$r" ]; then timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg; else DFVO_F16S_RUN=$r timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg; fi ) 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('frames/s', d['value'], 'steady', d['steady_state']['value'], '| frac', r['frac'], 'avg us', r['avg_launch_us'], 'conv family ms', r['conv_family_ms_per_pair'])
for c in r['by_config'][:3]: print('   ', c['kernel'][:60], c['ms_per_pair'], c['launches_per_pair'], c['tflops'])" >> $O
done
cat $O
