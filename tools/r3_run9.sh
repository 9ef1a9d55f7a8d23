#!/bin/bash
# round 3, GPU call 9: hardware-queue probe (which streams share a dispatch pipe?) and the one-round-trip ring of the
# K-sliced GEMM kernel (A/B against the three-stage ring)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd tools/ubench; for pre in 0 1 2; do GPU_MAX_HW_QUEUES=12 timeout 120 ./queue_probe $pre 12; done) 2>&1 | tee gpurun_out/r3i_queue_probe.txt | tail -45
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv" 2>&1 | grep -v "amdgpu.ids" | tail -2
for pf in 8 3 8 3; do
  DFVO_F16G_PF=$pf timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('PF=$pf', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'family ms', r['conv_family_ms_per_pair'], [ (k['kernel'][:22],k['ms_per_pair'],k['launches_per_pair']) for k in r['by_config'][:3]])"
done | tee gpurun_out/r3i_pf_ab.txt
