#!/bin/bash
# round 3, GPU call 37: the window layers on random / zero / constant operands (same instruction stream): how much of the launch time is the power limit
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for m in random zero_x zero_all const random; do
  echo "== operands $m"; OPERANDS=$m DFVO_F16S_V2=1 timeout 300 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3aj_operands.txt
