#!/bin/bash
# round 3, GPU call 34: flow-net instances 1 / 2 / 3 on the pipe-aware build; the K-over-workgroups conv cases (child pytest)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "k_divided" 2>&1 | tail -2
for rep in 1 2; do for n in 2 3 1; do
  DFVO_FLOW_INSTANCES=$n timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flow instances $n', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done; done
timeout 200 python tools/rccl_smoke.py > /tmp/rccl.txt 2>&1; grep -c "rccl smoke ok" /tmp/rccl.txt; grep "rccl smoke ok" /tmp/rccl.txt
} | tee gpurun_out/r3ag_instances.txt
