#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_pnp_gpu.py tests/test_e2e_gpu.py -q -m gpu -x > $O/e2e.log 2>&1; echo "pnp+e2e rc $?"; tail -3 $O/e2e.log
DFVO_TRACK_TRACE=1 DFVO_BENCH_TRACE=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -a "host ms\|track host" | tail -3
python bench.py --steps 30 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --height 960 --width 1280 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 rc $?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --height 1280 --width 1920 --e-max-iters 8192 --kp-bestn 20000 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 rc $?"; tail -2 $O/bench_cfg5.err
python - <<PY
import json
for f in ("bench_default","bench_cfg4","bench_cfg5"):
    try:
        d=json.load(open("$O/"+f+".json")); r=d["roofline"]
        print(f, d["value"], "fps", d["ms_per_step"], "ms | E/PnP/const", d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"], d["config"]["constant_motion"], "| fam", r["conv_family_achieved"], "TF/s", r["conv_family_ms_per_pair"], "ms | GF", r["algorithmic_gflop_per_pair"], "| dom", r["achieved"], r["frac"])
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_trace -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 3 > /tmp/b0.log 2>&1
echo "trace rc $?"; f=$(find /tmp/p_trace -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/$O/r2b_rocprofv3_kernel_stats.csv; tail -1 /tmp/b0.log | cut -c1-150
