#!/bin/bash
# final verification on the GPU box: full -m gpu suite and smoke()
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/final_tests.log
cat gpurun_out/final_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/final_smoke.log
cat gpurun_out/final_smoke.log
