#!/bin/bash
# BASELINE configs 4 and 5 through bench.py (records for profiles/)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 400 python bench.py --no-cpu-baseline --height 960 --width 1280 --steps 10 --warmup 3 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
timeout 400 python bench.py --no-cpu-baseline --height 1280 --width 1920 --e-max-iters 8192 --kp-bestn 20000 --steps 10 --warmup 3 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
for f in gpurun_out/bench_cfg4.json gpurun_out/bench_cfg5.json; do python -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['conv_family_achieved'])"; done
