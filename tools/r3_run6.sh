#!/bin/bash
# round 3, GPU call 6: batched epilogue (bias hoisted, stores back to back) -- parity, per-layer A/B of the two window
# skeletons, bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee gpurun_out/r3f_ops.log
DFVO_F16S_V2=1 timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee -a gpurun_out/r3f_ops.log
for v in 0 1; do
  DFVO_F16S_V2=$v timeout 300 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3f_layers_v$v.txt
done
paste gpurun_out/r3f_layers_v0.txt gpurun_out/r3f_layers_v1.txt | awk '{print $1,$2,"| v0",$5,$7,$10,"| v1",$15,$17,$20}'
for v in 0 1; do
  DFVO_F16S_V2=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3f_bench_v$v.json 2> gpurun_out/r3f_bench_v$v.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3f_bench_v$v.json").read().strip().splitlines()[-1])
    print("V2=$v", d["value"], d["ms_per_step"], "steady", d["steady_state"], "exact", d["exact_fp32"], "frac", d["roofline"]["frac"], d["roofline"]["conv_family_ms_per_pair"], d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"])
    for k in d["roofline"]["by_config"][:4]: print("   ", k["kernel"][:40], k["ms_per_pair"], k["launches_per_pair"], k["tflops"])
except Exception as e:
    print("failed", e, open("gpurun_out/r3f_bench_v$v.err").read()[-2500:])
PY
done
