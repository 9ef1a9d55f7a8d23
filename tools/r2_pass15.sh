#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_e2e_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/pass15_tests.log
cat gpurun_out/pass15_tests.log
timeout 600 python bench.py --steps 60 --warmup 10 > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err
tail -c 600 gpurun_out/bench_r2c.json
