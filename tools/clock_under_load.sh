#!/bin/bash
# shader clock / power while the nets run back to back (DFVO_CONV_PRECISION as exported by the caller)
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4
(STEPS=${STEPS:-4000} python tools/bench_nets_only.py > /tmp/bn.log 2>&1) &
BP=$!
sleep ${WARM:-25}
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
tail -1 /tmp/bn.log
