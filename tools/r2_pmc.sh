#!/bin/bash
# hardware-counter passes over bench.py (separate runs per counter, as MI355X_MICROARCH.md prescribes; never combined with
# the sys/hip/hsa trace domains) -> gpurun_out/r2b_pmc_bench.json, gpurun_out/r2b_mfma_busy.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o r -- $CMD > /tmp/b_$c.log 2>&1
  f=$(find /tmp/p_$c -name "*counter_collection.csv" | head -1)
  echo "$c -> $f $(wc -l < $f)"
  cp $f /tmp/$c.csv
done
python $R/tools/pmc_traffic.py /tmp/FETCH_SIZE.csv /tmp/WRITE_SIZE.csv $R/gpurun_out/r2b_pmc_bench.json | tail -10
python - <<'PY'
import csv,collections,re,os
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('/tmp/SQ_VALU_MFMA_BUSY_CYCLES.csv')):
    if r['Counter_Name']!='SQ_VALU_MFMA_BUSY_CYCLES': continue
    m=re.search(r"(conv_\w+_kernel<[^>]*>)", r['Kernel_Name'])
    if not m: continue
    a=acc[m.group(1)]; a[0]+=1; a[1]+=float(r['Counter_Value'])
with open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2b_mfma_busy.txt','w') as f:
    for k,(n,s) in sorted(acc.items(), key=lambda kv:-kv[1][1]):
        line="%-45s dispatches %5d  SQ_VALU_MFMA_BUSY_CYCLES/dispatch %.4g"%(k,n,s/n)
        print(line); f.write(line+"\n")
PY
