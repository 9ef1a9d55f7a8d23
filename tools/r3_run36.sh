#!/bin/bash
# round 3, GPU call 36: guard test of track_begin, a long run of the default workload (200 pairs), BASELINE configs 4 and 5 on this build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -x -k "refuses or carry" 2>&1 | tail -2
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r3_bench_200steps.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench_200steps.json').read().strip().splitlines()[-1]); print('200 steps:', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'recomputed', d['features_recomputed']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
timeout 400 python bench.py --no-cpu-baseline --height 960 --width 1280 --steps 10 --warmup 3 > gpurun_out/r3_bench_config4_1280x960.json 2> gpurun_out/bench_cfg4.err
timeout 600 python bench.py --no-cpu-baseline --height 1280 --width 1920 --e-max-iters 8192 --kp-bestn 20000 --steps 10 --warmup 3 > gpurun_out/r3_bench_config5_1920x1280_8192hyp_20kkp.json 2> gpurun_out/bench_cfg5.err
for f in gpurun_out/r3_bench_config4_1280x960.json gpurun_out/r3_bench_config5_1920x1280_8192hyp_20kkp.json; do python -c "
import sys,json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], 'exact', d['exact_fp32'] and d['exact_fp32']['value'], r['achieved'], r['frac'], r['conv_family_achieved'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])
except Exception as e: print('$f failed', e)"; done
} | tee gpurun_out/r3ai_long_and_configs.txt
