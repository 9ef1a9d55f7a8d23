#!/bin/bash
# round 4: full -m gpu suite on the cleaned build, default bench, config-3 job, BASELINE.md section-4 CPU protocol
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r4k_tests.txt
tail -4 gpurun_out/r4k_tests.txt
timeout 600 python bench.py > gpurun_out/r4k_bench_default.json 2> gpurun_out/r4k_bench_default.err
tail -1 gpurun_out/r4k_bench_default.json | cut -c1-400
timeout 300 python bench.py --sequences kitti-lengths --scale 0.02 2>/dev/null | tail -1 > gpurun_out/r4k_bench_config3.json
cut -c1-200 gpurun_out/r4k_bench_config3.json
