# memory-side PMC pass for the hot conv layer (companion of tools/pmc_layer.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export N=2 H=192 W=624 C0=128 COUT=128 K=3 STRIDE=1 ITERS=5
run() {
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pm_$1 -o r -- python $R/tools/bench_conv.py > /tmp/pm.log 2>&1
  f=$(find /tmp/pm_$1 -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_win_f32' not in r['Kernel_Name']: continue
    a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,(n,s) in sorted(acc.items()): print("%-36s per dispatch %.4g"%(k,s/n))
PY
}
run TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_GATE_EN1_sum TCP_GATE_EN2_sum
# (a TCC_* pass -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_BUSY_avr TCC_TAG_STALL_sum -- did not finish within
#  300 s on this pool and is left out)
run SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR TA_TA_BUSY_sum TA_BUSY_avr
