#!/bin/bash
# round 3, GPU call 1: the new parity tests (config 5 through the fused pipeline, images->pose at 960x1280 / 1280x1920,
# forced PnP at KITTI size, from-images accounting) + baseline numbers and a per-layer conv table for the kernel work
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_trajectory_gpu.py -q -m gpu -x -s 2>&1 | grep -v "amdgpu.ids" | grep -E "FROM-IMAGES|config 5|passed|failed|Error|error|assert|pair [0-9]+:" | tail -80 > gpurun_out/r3a_tests.log
tail -30 gpurun_out/r3a_tests.log
rm -f gpurun_out/r3a_layers_f16x3.csv gpurun_out/r3a_layers_fp32.csv
DFVO_CONV_PROFILE_CSV=gpurun_out/r3a_layers_f16x3.csv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a_bench_f16x3.json 2> gpurun_out/r3a_bench_f16x3.err
DFVO_CONV_PROFILE_CSV=gpurun_out/r3a_layers_fp32.csv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv-precision fp32 > gpurun_out/r3a_bench_fp32.json 2> gpurun_out/r3a_bench_fp32.err
python - <<'PY'
import json
for m in ("f16x3","fp32"):
    try:
        d=json.loads(open("gpurun_out/r3a_bench_%s.json"%m).read().strip().splitlines()[-1])
        print(m, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_family_ms_per_pair"])
    except Exception as e:
        print(m, "failed", e)
PY
