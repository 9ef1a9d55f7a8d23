#!/bin/bash
# round 3, GPU call 18: same-box A/B: host-decided PnP (previous library) vs device-gated draws, chain_ahead 0 / 1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=df-vo_amd/lib
cp $L/libdfvo_hip.so /tmp/new.so
one() {
  DFVO_BENCH_CHAIN_AHEAD=$2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 chain_ahead $2', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
}
for rep in 1 2 3; do
  cp $L/libdfvo_hip_old.so $L/libdfvo_hip.so; one old 0
  cp /tmp/new.so $L/libdfvo_hip.so; one new 0; one new 1
done | tee gpurun_out/r3r_ab.txt
