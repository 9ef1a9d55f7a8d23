#!/bin/bash
# round 3, GPU call 28: register-tiled correlation kernel: parity, per-level timing vs the first kernel, nets tests, pipeline A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -s -k correlation 2>&1 | grep -E "correlation|bit-exact|passed|failed|Error" | tail -16
for rt in 0 1; do DFVO_CORR_RT=$rt timeout 200 python tools/bench_corr.py 2>&1 | grep -v amdgpu.ids; done
timeout 900 python -m pytest tests/test_nets_gpu.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do for rt in 0 1; do
  DFVO_CORR_RT=$rt timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('corr rt $rt', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done; done
} | tee gpurun_out/r3aa_corr.txt
