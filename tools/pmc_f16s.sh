#!/bin/bash
# SQ counters of the f16x3 window kernel on one layer (separate passes; counter collection only, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONLY="${ONLY:-L2 128->128}"
CMD="python $R/tools/bench_f16s.py"
OUT=$R/gpurun_out/pmc_f16s.txt
: > $OUT
pass() {
  local tag=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pf_$tag -o r -- $CMD > /tmp/pf_$tag.log 2>&1
  f=$(find /tmp/pf_$tag -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $tag ($*): no output: $(tail -3 /tmp/pf_$tag.log | tr '\n' ' ')" >> $OUT; return; fi
  python - "$f" >> $OUT <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_win_f16s' not in r['Kernel_Name']: continue
    a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,(n,s) in sorted(acc.items()): print("%-32s per dispatch %.5g (n=%d)"%(k,s/n,n))
PY
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass b SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM
pass d SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
cat $OUT
