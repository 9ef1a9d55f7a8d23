#!/bin/bash
# round 3, GPU call 4: why is the one-wave-per-SIMD skeleton slower?  ablations (same instruction stream, loads redirected
# to cached addresses) and SQ counters on the level-2 128->128 layer
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
sed -i 's/^LAYERS = \[/LAYERS_ALL = [/' tools/bench_f16s_v2.py
sed -i 's/^g = torch.Generator/LAYERS = [l for l in LAYERS_ALL if l[0] in os.environ.get("ONLY", "L2 128->128,L2 128->64,c5 128->128").split(",")]\ng = torch.Generator/' tools/bench_f16s_v2.py
for abl in 0 1 2 3; do
  echo "== V2=1 ABL=$abl"; DFVO_F16S_V2=1 DFVO_F16S2_ABL=$abl timeout 200 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids | head -3
done | tee gpurun_out/r3d_ablation.txt
echo "== V2=0"; DFVO_F16S_V2=0 timeout 200 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids | head -3 | tee -a gpurun_out/r3d_ablation.txt
cd /tmp
pass() {
  local v=$1 tag=$2; shift; shift
  DFVO_F16S_V2=$v ONLY="L2 128->128" timeout 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pf_${v}_$tag -o r -- python $GRAFT_REPO_ROOT/tools/bench_f16s_v2.py > /tmp/pf_${v}_$tag.log 2>&1
  f=$(find /tmp/pf_${v}_$tag -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $v $tag: no output: $(tail -3 /tmp/pf_${v}_$tag.log | tr '\n' ' ')"; return; fi
  python - "$f" "$v" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv_win_f16s' not in r['Kernel_Name']: continue
    a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,(n,s) in sorted(acc.items()): print("V2=%s %-32s per dispatch %.5g (n=%d)"%(sys.argv[2],k,s/n,n))
PY
}
for v in 1 0; do
pass $v a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass $v b SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
pass $v c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM
pass $v d SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r3d_pmc_v2_vs_v0.txt
