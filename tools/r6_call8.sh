#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for v in 0 1; do
DFVO_REG_HEAD_LDS=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ms$v -o r -- python $R/bench.py --surface mirrors --steps 30 --warmup 3 --no-other-legs > /tmp/ms.log 2>&1
f=$(find /tmp/p_ms$v -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r6i_mirrors_kernel_stats_lds$v.csv
echo "== DFVO_REG_HEAD_LDS=$v"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_reg_head", "k_warp", "k_deconv", "k_flow_mean", "k_correlation", "k_reg_prep", "k_pool", "k_kp_", "fillBuffer", "k_maxpool", "conv_head", "conv_taps")):
        print("%-60s calls %5s avg %9.1f us total %8.2f ms (%.2f%%)" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.1f ms" % (tot / 1e6))
PY
done
