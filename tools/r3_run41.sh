#!/bin/bash
# round 3, GPU call 41: level-2 moduleFeat convolutions fused into one launch: bit-identity A/B, nets / pipeline tests, rate A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_nets_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_e2e_gpu.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do for f in 0 1; do
  DFVO_FLOW_FUSE_FEAT=$f timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse_feat $f', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done; done
} | tee gpurun_out/r3am_fuse_feat.txt
