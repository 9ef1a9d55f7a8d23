#!/bin/bash
# profiles of the default bench command (TAG=r4 bash tools/profile.sh on the GPU box): native rocprofv3 kernel-trace statistics, then hardware-counter passes
# (one counter per run; counter collection is never combined with the sys/hip/hsa trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r6}
mkdir -p $R/gpurun_out
CMD="python $R/bench.py --no-cpu-baseline --no-roofline --no-exact-leg --no-other-legs --steps 20 --warmup 5"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o r -- $CMD > /tmp/b_stats.log 2>&1
f=$(find /tmp/p_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o r -- $CMD > /tmp/b_$c.log 2>&1
  f=$(find /tmp/p_$c -name "*counter_collection.csv" | head -1)
  echo "$c -> $f $(wc -l < $f)"
  cp $f /tmp/$c.csv
done
python $R/tools/pmc_traffic.py /tmp/FETCH_SIZE.csv /tmp/WRITE_SIZE.csv $R/gpurun_out/${TAG}_pmc_bench.json | tail -6
python - <<'PY'
import csv,collections,re,os
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('/tmp/SQ_VALU_MFMA_BUSY_CYCLES.csv')):
    if r['Counter_Name']!='SQ_VALU_MFMA_BUSY_CYCLES': continue
    m=re.search(r"(conv_\w+_kernel<[^>]*>)", r['Kernel_Name'])
    if not m: continue
    a=acc[m.group(1)]; a[0]+=1; a[1]+=float(r['Counter_Value'])
with open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/%s_mfma_busy.txt' % os.environ.get('TAG','r5'),'w') as f:
    for k,(n,s) in sorted(acc.items(), key=lambda kv:-kv[1][1]):
        line="%-45s dispatches %5d  SQ_VALU_MFMA_BUSY_CYCLES/dispatch %.4g"%(k,n,s/n)
        print(line); f.write(line+"\n")
PY
head -40 $R/gpurun_out/${TAG}_rocprofv3_kernel_stats.csv | cut -c1-160
