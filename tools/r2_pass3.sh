#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "f16x3" -x > $O/f16_test.log 2>&1; echo "f16 tests rc $?"; tail -15 $O/f16_test.log
python tools/bench_f16s.py > $O/f16_bench.log 2>&1; echo "bench rc $?"; cat $O/f16_bench.log | tail -14
for m in fp32 f16x3; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --solver-inputs synthetic --conv-precision $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('syn $m', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['conv_family_ms_per_pair'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --conv-precision $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nets $m', d['value'], d['ms_per_step'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])"
done
