#!/bin/bash
# kernel timeline of the solver-side kernels over one steady-state pair period of bench.py (nets running concurrently)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o r -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 5 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/solver_timeline_bench.txt <<'PY'
import csv,sys,re
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sh=[i for i,r in enumerate(rows) if 'k_mt_shuffle_all' in r['Kernel_Name']]
ts=[int(rows[i]['Start_Timestamp']) for i in sh]
print("shuffle periods (ms):", [round((b-a)/1e6,2) for a,b in zip(ts,ts[1:])])
a,b=sh[24],sh[25]
t0=int(rows[a]['Start_Timestamp'])
nets={}
prev_end=None
for r in rows[a:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=re.sub(r'\(.*','',r['Kernel_Name']).replace('dfvo::','').replace('void ','')
    q=r.get('Queue_Id','?')
    if name.startswith('conv_') or name in ('k_correlation','k_warp','k_deconv_dw','k_reg_head','k_flow_mean','k_resize_bilinear','k_lanczos_pass','k_reg_prep','k_flow_post','k_u8_to_net','k_maxpool','k_upsample') :
        nets.setdefault(q,[0,0]); nets[q][0]+=1; nets[q][1]+=e-s
        continue
    print("%8.1f us  dur %7.1f  q%-3s %s"%((s-t0)/1e3,(e-s)/1e3,q,name))
print("net-side kernels by queue in this window:")
for q,(n,d) in nets.items(): print("  q%s: %d kernels, busy %.1f us"%(q,n,d/1e3))
PY
head -120 $R/gpurun_out/solver_timeline_bench.txt
