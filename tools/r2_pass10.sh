#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_pnp_gpu.py -q -m gpu -x -k "f16x3 or coplanar" > $O/f16_test.log 2>&1; echo "tests rc $?"; tail -3 $O/f16_test.log
echo "--- 8-wave tiles on"; python tools/bench_f16s.py 2>/dev/null | tail -11
echo "--- 8-wave tiles off"; DFVO_F16S_FILL8=100000000 python tools/bench_f16s.py 2>/dev/null | tail -11
for v in 400 100000000; do
DFVO_F16S_FILL8=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('fill8=$v', d['value'], d['ms_per_step'], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], 'fam', r['conv_family_achieved'], r['conv_family_ms_per_pair'], 'dom', r['achieved'], r['avg_launch_us'])"
done
