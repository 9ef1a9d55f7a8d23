#!/bin/bash
# round 3, last GPU call: smoke + nets / pipeline / e2e tests on the final build (moduleFeat fusion off by default)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_nets_gpu.py tests/test_pipeline_gpu.py tests/test_e2e_gpu.py tests/test_trajectory_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rate', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
} | tee gpurun_out/r3an_last.txt
