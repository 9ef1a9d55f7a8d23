#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python tools/bench_f16s.py 2>&1 | grep -v amdgpu.ids | cut -c1-140 > gpurun_out/f16_spread.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "f16x3" 2>&1 | tail -2 >> gpurun_out/f16_spread.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['by_config'][0]['ms_per_pair'])" >> gpurun_out/f16_spread.log; done
cat gpurun_out/f16_spread.log
