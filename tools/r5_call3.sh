#!/bin/bash
# round 5, GPU call 3: session with pipe-aware streams, f16 reports, ablation of launch families, the default line with its new legs
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r5c
( timeout 900 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x -s 2>&1 | grep -E "session|passed|failed|FAILED|Error|assert" | tail -12 ) > ${O}_dropin.txt; tail -3 ${O}_dropin.txt
for cfgv in "0 f16x3" "1 f16x3" "1 f16x3" "1 fp32"; do set -- $cfgv
  DFVO_SESSION=$1 timeout 300 python bench.py --surface mirrors --steps 20 --warmup 3 --conv-precision $2 2>/dev/null | tail -1 > ${O}_mirrors_s$1_$2.json
  python -c "
import json; d=json.loads(open('${O}_mirrors_s$1_$2.json').read())
print('mirrors session=$1 $2: frames/s', d['value'], d['stage_ms_per_pair'], d.get('session'))"
done 2>&1 | tee ${O}_mirrors.txt
( timeout 900 python -m pytest tests/test_f16_mode_gpu.py -m gpu -q -s 2>&1 | grep -E "F16-MODE|passed|failed|FAILED|Error|assert" | tail -20 ) > ${O}_f16_mode.txt; cat ${O}_f16_mode.txt | cut -c1-400
B="python bench.py --solver-inputs synthetic --steps 40 --warmup 8 --no-cpu-baseline --no-exact-leg --no-other-legs --no-roofline"
for rep in 1 2; do for ab in "" k s h ks ksh w; do
  echo "ABLATE='$ab' rep $rep $(DFVO_ABLATE=$ab timeout 200 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_state']['value'])")"
done; done 2>&1 | tee ${O}_ablation.txt
S=$(date +%s); timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json 2> ${O}_bench_default.err; echo "default bench wall $(( $(date +%s) - S )) s"
tail -3 ${O}_bench_default.err
python -c "
import json; d=json.loads(open('${O}_bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'frac', d['roofline']['frac'])
print('hbm', {k:v for k,v in d['roofline'].get('hbm',{}).items() if k!='other_kernels'})
print('dropin', d['dropin_surface'])
for k,v in (d['other_configs'] or {}).items(): print(k, v)
"
