#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_e2e_gpu.py tests/test_trajectory_gpu.py -q -m gpu -x > $O/e2e.log 2>&1; echo "e2e+traj rc $?"; tail -3 $O/e2e.log
DFVO_CONV_PRECISION=f16x3 python tools/bench_stages.py 2>/dev/null | tail -4
DFVO_TRACK_TRACE=1 DFVO_BENCH_TRACE=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -a "host ms\|track host\|\"value\"" | cut -c1-220 | tail -4
python bench.py --steps 30 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('default', d['value'], 'fps', d['ms_per_step'], 'ms | E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'], '| fam', r['conv_family_achieved'], r['conv_family_ms_per_pair'], '| cpu', d['cpu_baseline']['value'])"
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > /tmp/b0.log 2>&1
echo "trace rc $?"; f=$(find /tmp/p_trace -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/$O/r2b_rocprofv3_kernel_stats.csv; tail -1 /tmp/b0.log | cut -c1-150
