#!/bin/bash
# round 4: the exact-fp32 register-ring kernel (conv_gemm_f32g): op / net parity, then the fp32-mode pair rate with
# DFVO_F32G = 0 (round 3's path), 1 (small maps), 2 (every non-window layer)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4e_tests.txt
cat gpurun_out/r4e_tests.txt
O=gpurun_out/r4e_fp32_f32g_ab.txt; : > $O
for v in 0 1 2 0 1 2; do
  echo "== DFVO_F32G=$v" >> $O
  DFVO_F32G=$v timeout 300 python bench.py --conv-precision fp32 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('frames/s', d['value'], 'steady', d['steady_state']['value'], 'conv family ms', r['conv_family_ms_per_pair'])
for c in r['by_config'][:7]: print('   ', c['kernel'][:70], c['ms_per_pair'], c['launches_per_pair'], c['tflops'])" >> $O
done
cat $O
