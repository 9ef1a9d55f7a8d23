#!/bin/bash
# SQ counters of ONE convolution layer (tools/bench_conv.py) under rocprofv3 --pmc, two passes of eight counters.
# usage: LAYER="N=1 H=352 W=1216 C0=3 COUT=32 K=7" [DFVO_TAPS=0] bash tools/pmc_layer.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
export DFVO_CONV_PRECISION=${DFVO_CONV_PRECISION:-f16x3} ITERS=10
for kv in $LAYER; do export $kv; done
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU"
P3="GRBM_GUI_ACTIVE"   # / the launch's wall time = the effective shader clock (MI355X_MICROARCH.md, DVFS give-back)
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/pl_$i
  timeout 200 rocprofv3 --pmc $P --output-format csv -d /tmp/pl_$i -o r -- python $R/tools/bench_conv.py > /tmp/pl_$i.log 2>&1
  grep "^conv" /tmp/pl_$i.log
  f=$(find /tmp/pl_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$TAG" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-48:]
    if "conv_" not in k: continue
    a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(sys.argv[2], k, {c: "%.3g" % (v[1] / v[0]) for c, v in d.items()})
PY
done
