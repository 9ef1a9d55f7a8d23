#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -q -m gpu -x -rs > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('default', d['value'], d['ms_per_step'], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], 'fam', r['conv_family_achieved'], r['conv_family_ms_per_pair'], 'dom', r['achieved'])"
bash tools/r2_pmc.sh 2>&1 | tail -16
