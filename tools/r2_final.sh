#!/bin/bash
# round-2 final verification on the GPU box: full -m gpu suite, smoke(), bench (split-K finish A/B, then the default line)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/final_tests.log
cat gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/final_smoke.log
cat gpurun_out/final_smoke.log
for f in 1 0; do echo -n "DFVO_SPLITK_FUSED=$f " >> gpurun_out/final_tests.log; DFVO_SPLITK_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['conv_family_ms_per_pair'], r['conv_family_achieved'])" >> gpurun_out/final_tests.log; done
tail -2 gpurun_out/final_tests.log
timeout 600 python bench.py > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err
cut -c1-330 gpurun_out/bench_r2d.json
