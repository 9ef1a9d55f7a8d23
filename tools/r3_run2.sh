#!/bin/bash
# round 3, GPU call 2: the generic f16x3 GEMM kernel (conv_gemm_f16s.h) -- operator parity in both precisions, the nets,
# then bench A/B (DFVO_F16G=0 restores the fp32 kernels for those layers) with per-layer tables
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r3b_ops.log
tail -5 gpurun_out/r3b_ops.log
timeout 900 python -m pytest tests/test_nets_gpu.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r3b_nets.log
tail -5 gpurun_out/r3b_nets.log
for g in 1 0; do
  rm -f gpurun_out/r3b_layers_g$g.csv
  DFVO_F16G=$g DFVO_CONV_PROFILE_CSV=gpurun_out/r3b_layers_g$g.csv timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3b_bench_g$g.json 2> gpurun_out/r3b_bench_g$g.err
done
python - <<'PY'
import json
for g in (1,0):
    try:
        d=json.loads(open("gpurun_out/r3b_bench_g%d.json"%g).read().strip().splitlines()[-1])
        print("F16G=%d"%g, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_family_ms_per_pair"], d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"])
        for k in d["roofline"]["by_config"][:8]: print("   ", k)
    except Exception as e:
        print(g, "failed", e, open("gpurun_out/r3b_bench_g%d.err"%g).read()[-1500:])
PY
