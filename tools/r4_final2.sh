#!/bin/bash
# final verification of the round on the GPU box: whole -m gpu suite (anchor / from-images lines kept), error tools, profiles, bench
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "ANCHOR|FROM-IMAGES|passed|failed|FAILED|Error" > gpurun_out/r4x_tests.txt
tail -3 gpurun_out/r4x_tests.txt
for prec in fp32 f16x3; do
  DFVO_CONV_PRECISION=$prec timeout 300 python tools/flow_error_by_level.py 2>&1 | grep -v amdgpu > gpurun_out/r4x_levels_tunnel_$prec.txt
done
DFVO_CONV_PRECISION=fp32 timeout 300 python tools/flow_op_replay.py 2>&1 | grep -v amdgpu > gpurun_out/r4x_replay_tunnel_fp32.txt
TAG=r4 bash tools/profile.sh > gpurun_out/r4x_profile.log 2>&1
cd $R
timeout 400 python bench.py > gpurun_out/r4x_bench_default.json 2> gpurun_out/r4x_bench_default.err
tail -c 600 gpurun_out/r4x_bench_default.json
