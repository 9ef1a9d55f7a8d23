#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6c_session_order.txt; : > $O
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>gpurun_out/r6c_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lbl', d['value'], d['stage_ms_per_pair'])" >> $O
  grep "session trace" gpurun_out/r6c_err.txt | tail -4 >> $O
  grep "stream pool" gpurun_out/r6c_err.txt | tail -2 >> $O
}
run base_early0 DFVO_SESSION_POSE_EARLY=0 DFVO_SESSION_TRACE=1 DFVO_STREAM_PROBE_VERBOSE=1
run early1 DFVO_SESSION_POSE_EARLY=1 DFVO_SESSION_TRACE=1
run order1_depth_first DFVO_SESSION_ORDER=1 DFVO_SESSION_TRACE=1
run order2_serial_depth_then_flow DFVO_SESSION_ORDER=2 DFVO_SESSION_TRACE=1
run order3_serial_flow_then_depth DFVO_SESSION_ORDER=3 DFVO_SESSION_TRACE=1
run early1_notrace DFVO_SESSION_POSE_EARLY=1
run order2_notrace DFVO_SESSION_ORDER=2
cat $O
