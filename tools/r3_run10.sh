#!/bin/bash
# round 3, GPU call 10: pipe-aware stream assignment (probe on / off x torch first / not), streaming-shape A/B, conv tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | grep -v "amdgpu.ids" | tail -12
for cfg in "1 0" "1 1" "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  DFVO_STREAM_PROBE=$1 DFVO_STREAM_PROBE_VERBOSE=1 DFVO_BENCH_TORCH_FIRST=$([ $2 = 1 ] && echo 1) timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe=$1 torch_first=$2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
  grep "stream pool" /tmp/err.txt | head -1
done | tee gpurun_out/r3j_stream_probe_ab.txt
for tc1 in 0 1 0 1; do
  DFVO_F16G_STREAM_TC1=$tc1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('STREAM_TC1=$tc1', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'family ms', r['conv_family_ms_per_pair'], [ (k['kernel'][:22],k['ms_per_pair'],k['launches_per_pair']) for k in r['by_config'][:3]])"
done | tee gpurun_out/r3j_stream_tc1_ab.txt
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_pipeline_gpu.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -3
