#!/bin/bash
# per-kernel device timeline of ONE frame of the class-surface loop (bench.py --surface mirrors): start offset, duration, stream,
# queue, grid, registers of every kernel, in start order (rocprofv3 kernel trace; GPU box).  TAG names the output.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r6a}
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o r -- python $R/bench.py --surface mirrors --steps 8 --warmup 3 --no-other-legs > /tmp/tl.log 2>&1
f=$(find /tmp/p_tl -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/${TAG}_timeline_detail.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def name(r): return r["Kernel_Name"].replace("void ", "").replace("dfvo::", "")[:58]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "k_img_u8_to_flow_input" in r["Kernel_Name"]]
fi = len(starts) - 3
a, b = starts[fi], starts[fi + 1]
t0 = rows[a]["s"]
sk = "Stream_Id" if "Stream_Id" in rows[0] else None
print("frame %d of %d; columns: start_us dur_us stream queue grid wg vgpr agpr lds name" % (fi, len(starts)))
for r in rows[a:b]:
    print("%9.1f %7.1f  s%-3s q%-3s g%-7s w%-4s v%-3s a%-3s l%-6s %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r.get(sk, "?") if sk else "?",
          r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], name(r)))
PY
wc -l $R/gpurun_out/${TAG}_timeline_detail.txt
