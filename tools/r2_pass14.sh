#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "f16x3" > $O/f16_test.log 2>&1; echo "tests rc $?"; tail -2 $O/f16_test.log; grep -a "^E  " $O/f16_test.log | head -5
echo "--- ws on"; timeout 120 python tools/bench_f16s.py 2>/dev/null | tail -11 | cut -c1-17,60-150
echo "--- ws off, variant 1"; DFVO_F16S_WS=0 DFVO_F16S_VARIANT=1 DFVO_F16S_FILL8=100000000 python tools/bench_f16s.py 2>/dev/null | tail -11 | cut -c1-17,60-150
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench ws', d['value'], d['ms_per_step'], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], 'fam', r['conv_family_achieved'], r['conv_family_ms_per_pair'], 'dom', r['achieved'], r['avg_launch_us'])"
