cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/bench_conv.py"
export N=2 H=192 W=624 C0=128 COUT=128 K=3 STRIDE=1 ITERS=5
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/p1 -o r -- $CMD > /tmp/l1.log 2>&1
f=$(find /tmp/p1 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_win_f32' not in r['Kernel_Name']: continue
    a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,(n,s) in sorted(acc.items()): print("%-24s per dispatch %.4g (n=%d)"%(k,s/n,n))
PY
tail -2 /tmp/l1.log
