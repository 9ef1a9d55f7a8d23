#!/bin/bash
# round 3, GPU call 21: window kernel, requests fenced into the MFMA shadow (F16S2_LATE 1) vs ahead of the tap (0): per layer + pipeline, same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=df-vo_amd/lib
cp $L/libdfvo_hip.so /tmp/late1.so
bench() {
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
}
{
for v in late0 late1; do
  if [ $v = late0 ]; then cp $L/libdfvo_hip_late0.so $L/libdfvo_hip.so; else cp /tmp/late1.so $L/libdfvo_hip.so; fi
  echo "== $v"; DFVO_F16S_V2=1 timeout 300 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2 3; do
  cp $L/libdfvo_hip_late0.so $L/libdfvo_hip.so; bench late0
  cp /tmp/late1.so $L/libdfvo_hip.so; bench late1
done
} | tee gpurun_out/r3u_late_requests.txt
cp /tmp/late1.so $L/libdfvo_hip.so
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k conv 2>&1 | tail -2 | tee -a gpurun_out/r3u_late_requests.txt
