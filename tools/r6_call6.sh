#!/bin/bash
# kernel statistics of the class-surface loop (rocprofv3 --kernel-trace --stats), and the PnP pair's per-kernel timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ms -o r -- python $R/bench.py --surface mirrors --steps 30 --warmup 3 --no-other-legs > /tmp/ms.log 2>&1
f=$(find /tmp/p_ms -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r6g_mirrors_kernel_stats.csv
head -45 $f | cut -c1-150
t=$(find /tmp/p_ms -name "*kernel_trace.csv" | head -1)
python - "$t" > $R/gpurun_out/r6g_pnp_pair_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "k_img_u8_to_flow_input" in r["Kernel_Name"]]
# the last frame that contains a k_pnp kernel
sel = None
for fi in range(len(starts) - 1):
    if any("k_pnp" in r["Kernel_Name"] for r in rows[starts[fi]:starts[fi + 1]]):
        sel = fi
a, b = starts[sel], starts[sel + 1]
t0 = rows[a]["s"]
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void ", "").replace("dfvo::", "")
    if "conv_" in n or "k_warp" in n or "k_corr" in n or "resize" in n: continue
    print("%9.1f %7.1f  s%-3s g%-7s w%-4s %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r.get("Stream_Id", "?"), r["Grid_Size_X"], r["Workgroup_Size_X"], n[:70]))
PY
tail -70 $R/gpurun_out/r6g_pnp_pair_timeline.txt
