#!/bin/bash
# round 3, GPU call 27: split-K fused finish under pipeline load (A/B in subprocesses)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nets_gpu.py -q -m gpu -x -k "splitk" 2>&1 | tail -15 | tee gpurun_out/r3z_splitk.txt
