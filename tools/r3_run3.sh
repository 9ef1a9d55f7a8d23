#!/bin/bash
# round 3, GPU call 3: the one-wave-per-SIMD window skeleton (conv_win_f16s2.h) -- parity, per-layer A/B, bench A/B;
# K-sliced generic kernel with a deeper ring; rocprofv3 kernel trace of the default bench for real small-kernel durations
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
DFVO_F16S_V2=1 timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | grep -v "amdgpu.ids" | tail -4 > gpurun_out/r3c_ops_v2.log
tail -3 gpurun_out/r3c_ops_v2.log
DFVO_F16S_V2=3 timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "f16x3" 2>&1 | grep -v "amdgpu.ids" | tail -3 >> gpurun_out/r3c_ops_v2.log
tail -2 gpurun_out/r3c_ops_v2.log
for v in 0 1 3; do
  DFVO_F16S_V2=$v timeout 300 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c_layers_v$v.txt
  tail -1 gpurun_out/r3c_layers_v$v.txt
done
paste gpurun_out/r3c_layers_v0.txt gpurun_out/r3c_layers_v1.txt gpurun_out/r3c_layers_v3.txt | awk '{print $1,$2,"| v0",$5,$9,"| v1",$14,$18,"| v3",$23,$27}' | head -12
for cfg in "0 3" "1 3" "0 5"; do
  set -- $cfg
  DFVO_F16S_V2=$1 DFVO_F16G_PF=$2 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3c_bench_v$1_pf$2.json 2> gpurun_out/r3c_bench_v$1_pf$2.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3c_bench_v$1_pf$2.json").read().strip().splitlines()[-1])
    print("V2=$1 PF=$2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_family_ms_per_pair"], d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"])
    for k in d["roofline"]["by_config"][:4]: print("   ", k["kernel"][:40], k["ms_per_pair"], k["launches_per_pair"], k["tflops"])
except Exception as e:
    print("failed", e, open("gpurun_out/r3c_bench_v$1_pf$2.err").read()[-1500:])
PY
done
cd /tmp && rm -rf /tmp/prof_r3c && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3c -o r3c -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/r3c_prof_bench.json" 2> /tmp/prof_err.txt
cd "$GRAFT_REPO_ROOT"
f=$(find /tmp/prof_r3c -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r3c_rocprofv3_kernel_stats.csv && head -30 gpurun_out/r3c_rocprofv3_kernel_stats.csv | cut -c1-150
