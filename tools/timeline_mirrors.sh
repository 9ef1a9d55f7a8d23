#!/bin/bash
# device timeline of the class-surface frame loop (bench.py --surface mirrors): per frame, when each stream's kernels start and end
# relative to the frame's first kernel, and how much of that span the stream's kernels cover (rocprofv3 kernel trace; GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r5p}
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o r -- python $R/bench.py --surface mirrors --steps 8 --warmup 3 --no-other-legs > /tmp/tl.log 2>&1
f=$(find /tmp/p_tl -name "*kernel_trace.csv" | head -1)
echo "trace: $f $(wc -l < $f)"
python - "$f" > $R/gpurun_out/${TAG}_mirrors_timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
def name(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dfvo::", "")[:44]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Queue_ID"
# frames: each push starts with the flow net's input kernel
starts = [i for i, r in enumerate(rows) if "k_img_u8_to_flow_input" in r["Kernel_Name"]]
# with carried pyramids one k_img_u8_to_flow_input per frame
print("frames found:", len(starts))
role = {}
for r in rows:
    n = r["Kernel_Name"]
    q = r[qkey]
    if "k_correlation_rt" in n: role[q] = "flow"
    elif "k_disp_to_depth" in n: role[q] = "depth"
    elif "k_h_refine" in n: role[q] = "kp+H (s_pre)"
    elif "k_e_poly" in n: role.setdefault(q, "E side (rep)")
    elif "k_cheirality" in n: role[q] = "tracker"
for fi in range(max(0, len(starts) - 4), len(starts) - 1):
    a, b = starts[fi], starts[fi + 1]
    fr = rows[a:b]
    t0 = fr[0]["s"]
    print("\nframe %d: %d kernels, span %.3f ms to the next frame's first kernel" % (fi, len(fr), (rows[b]["s"] - t0) / 1e6))
    per = collections.OrderedDict()
    for r in fr:
        d = per.setdefault(r[qkey], [r["s"], r["e"], 0, 0])
        d[0] = min(d[0], r["s"]); d[1] = max(d[1], r["e"]); d[2] += r["e"] - r["s"]; d[3] += 1
    for q, (s, e, busy, n) in per.items():
        print("   queue %-4s %-14s first start %7.3f  last end %7.3f  kernel time %6.3f ms in %4d kernels" % (q, role.get(q, "?"), (s - t0) / 1e6, (e - t0) / 1e6, busy / 1e6, n))
    # flow stream: phases by marker kernels
    fq = [q for q in per if role.get(q) == "flow"]
    if fq:
        fl = [r for r in fr if r[qkey] == fq[0]]
        corr = [r for r in fl if "k_correlation_rt" in r["Kernel_Name"]]
        marks = [("first kernel", fl[0]["s"])] + [("correlation %d" % i, c["s"]) for i, c in enumerate(corr)] + [("last end", fl[-1]["e"])]
        print("   flow stream: " + " | ".join("%s %.3f" % (m, (t - t0) / 1e6) for m, t in marks))
        gaps = sum(max(0, fl[i + 1]["s"] - fl[i]["e"]) for i in range(len(fl) - 1))
        print("   flow stream: gaps between consecutive kernels %.3f ms, kernel time %.3f ms" % (gaps / 1e6, sum(r["e"] - r["s"] for r in fl) / 1e6))
PY
tail -60 $R/gpurun_out/${TAG}_mirrors_timeline.txt | cut -c1-260
