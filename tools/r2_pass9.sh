#!/bin/bash
# round-2 GPU pass 9: the remaining 8f branches (flow_ratio, homo_ratio, abs_diff, 8-bit bilinear resize)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_resize_gpu.py tests/test_tracker_gpu.py tests/test_dropin_gpu.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/pass9_tests.log
cat gpurun_out/pass9_tests.log
