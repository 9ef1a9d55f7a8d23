#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
O=gpurun_out/r6j_batched_loads.txt; : > $O
timeout 200 python tools/crc_flow.py 2>/dev/null | tail -1 >> $O
timeout 200 python tools/crc_flow.py fp32 2>/dev/null | tail -1 >> $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
for i in 1 2; do
timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('mirrors', d['value'], d['stage_ms_per_pair'])" >> $O
done
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-exact-leg --no-other-legs 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('fused', d['value'], 'steady', d['steady_state']['value'], 'frac', r['frac'], 'hbm', r.get('hbm', {}).get('frac'))" >> $O
DFVO_SESSION_TRACE=1 timeout 300 python bench.py --surface mirrors --steps 10 --warmup 5 2>&1 | grep "session trace" | tail -3 >> $O
cat $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ms -o r -- python $R/bench.py --surface mirrors --steps 30 --warmup 3 --no-other-legs > /tmp/ms.log 2>&1
f=$(find /tmp/p_ms -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r6j_mirrors_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_correlation", "k_pool", "conv_taps", "k_warp")):
        print("%-60s calls %5s avg %9.1f us total %8.2f ms (%.2f%%)" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.1f ms" % (tot / 1e6))
PY
