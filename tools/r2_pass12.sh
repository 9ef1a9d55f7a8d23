#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "--- nt $v"; DFVO_F16S_NT=$v DFVO_F16S_VARIANT=0 DFVO_F16S_FILL8=100000000 python tools/bench_f16s.py 2>/dev/null | tail -11 | cut -c1-17,60-150
done
