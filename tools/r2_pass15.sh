#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_nets_gpu.py -q -m gpu -x -k "splitk or graph_replay or depthnet" 2>&1 | tail -3 > gpurun_out/pass16_tests.log
for f in 1 0 1 0; do echo -n "DFVO_SPLITK_FUSED=$f " >> gpurun_out/pass16_tests.log; DFVO_SPLITK_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['conv_family_ms_per_pair'], r['conv_family_achieved'])" >> gpurun_out/pass16_tests.log; done
cat gpurun_out/pass16_tests.log
