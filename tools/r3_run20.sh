#!/bin/bash
# round 3, GPU call 20: k_e_poly waves packed four to a compute unit (DFVO_E_POLY_BLOCK 256 vs 64), same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
one() {
  DFVO_E_POLY_BLOCK=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('poly block $1', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], '| recomputed', d['features_recomputed']['value'])"
}
for rep in 1 2 3; do one 64; one 256; one 128; done | tee gpurun_out/r3t_poly_block.txt
timeout 900 python -m pytest tests/test_solvers_gpu.py tests/test_tracker_gpu.py -q -m gpu -x 2>&1 | tail -2 | tee -a gpurun_out/r3t_poly_block.txt
