#!/bin/bash
# round 3, GPU call 33: K divided over workgroups on the small maps (conv_gemm_f16s gridDim.z): parity, conv family by config, pipeline A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k conv 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_nets_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -2
for nz in 0 -1; do
  DFVO_F16G_NZ=$nz timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('NZ=$nz', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'family ms', r['conv_family_ms_per_pair'])
for c in r['by_config'][:4]: print('    ', c['kernel'][:50], c['ms_per_pair'], c['launches_per_pair'], c['tflops'])"
done
for rep in 1 2 3; do for nz in 0 -1; do
  DFVO_F16G_NZ=$nz timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NZ=$nz', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done; done
} | tee gpurun_out/r3af_nz.txt
