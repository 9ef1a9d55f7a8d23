#!/bin/bash
# round 3, GPU call 26: dfvo_set_sklearn_compat / dfvo_ransac_regressor tests, the tracker / pipeline suites on the default again, smoke
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tracker_gpu.py -q -m gpu -x -s -k "sklearn or ransac_regressor" 2>&1 | grep -E "sklearn|seeds on|results with|passed|failed|Error|assert" | tail -20 | tee gpurun_out/r3y_sklearn.txt
timeout 1500 python -m pytest tests/test_tracker_gpu.py tests/test_pipeline_gpu.py tests/test_e2e_gpu.py tests/test_dropin_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee -a gpurun_out/r3y_sklearn.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r3y_sklearn.txt
