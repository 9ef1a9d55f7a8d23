#!/bin/bash
# kernel timeline of the solver-side kernels over one steady-state pair period of bench.py (nets running concurrently)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/solver_timeline_bench.txt <<'PY'
import csv,sys,re
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sh=[i for i,r in enumerate(rows) if 'k_mt_shuffle_all' in r['Kernel_Name']]
print("shuffle starts (ms):", [round((int(rows[i]['Start_Timestamp'])-int(rows[sh[0]]['Start_Timestamp']))/1e6,3) for i in sh])
a,b=sh[-4],sh[-2]
t0=int(rows[a]['Start_Timestamp'])
nets={}
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=re.sub(r'\(.*','',r['Kernel_Name']).replace('dfvo::','').replace('void ','')
    if name.startswith('conv_') or name in ('k_correlation','k_warp','k_deconv_dw','k_reg_head','k_flow_mean','k_resize_bilinear','k_lanczos_pass') or 'rocclr' in name and False:
        q=r.get('Queue_Id','?'); nets.setdefault(q,[0,0,None,None]); nets[q][0]+=1; nets[q][1]+=e-s
        nets[q][2]=s if nets[q][2] is None else nets[q][2]; nets[q][3]=e
        continue
    print("%8.1f us  dur %7.1f  q%-3s %s"%((s-t0)/1e3,(e-s)/1e3,r.get('Queue_Id','?'),name))
print("net-side kernels by queue in this window:")
for q,(n,d,s,e) in nets.items(): print("  q%s: %d kernels, busy %.1f us, first %.1f last %.1f us"%(q,n,d/1e3,(s-t0)/1e3,(e-t0)/1e3))
PY
head -150 $R/gpurun_out/solver_timeline_bench.txt
