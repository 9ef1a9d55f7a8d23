#!/bin/bash
# round 5, GPU call 2: the frame session under the drop-in classes, staged ramp, window stagger, f16-mode diagnosis, other legs
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r5b
( timeout 900 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x -s 2>&1 | grep -E "session|modes|passed|failed|FAILED|Error|assert" | tail -30 ) > ${O}_dropin.txt; tail -4 ${O}_dropin.txt
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -s -k "dynamic_range or small_maps_with_k" 2>&1 | grep -E "f16x3 range|passed|failed|FAILED|Error|assert" | tail -30 ) > ${O}_ops.txt; tail -3 ${O}_ops.txt
for a in "256 640 mux" "376 1241 pot" "1280 1920 mux"; do timeout 200 python tools/f16_mode_debug.py $a 2>/dev/null | grep -v amdgpu; done > ${O}_f16_debug.txt; cat ${O}_f16_debug.txt
for cfgv in "0 f16x3" "1 f16x3" "1 fp32" "1 f16"; do set -- $cfgv
  DFVO_SESSION=$1 timeout 300 python bench.py --surface mirrors --steps 20 --warmup 3 --conv-precision $2 2>/dev/null | tail -1 > ${O}_mirrors_s$1_$2.json
  python -c "
import json; d=json.loads(open('${O}_mirrors_s$1_$2.json').read())
print('mirrors session=$1 $2: frames/s', d['value'], d['stage_ms_per_pair'], d.get('session'))"
done 2>&1 | tee ${O}_mirrors.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-leg --no-other-legs --no-roofline"
for rep in 1 2 3; do for r in 3 1 2; do
  echo "RAMP=$r rep $rep $(DFVO_BENCH_RAMP=$r timeout 200 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'steady', d['steady_state']['value'], d['config']['tracked_by_E'])")"
done; done 2>&1 | tee ${O}_ramp_ab.txt
for st in 0 3 5; do
  DFVO_WIN_STAGGER=$st timeout 200 python tools/bench_window_layers.py 2>/dev/null | grep -v amdgpu > ${O}_layers_stagger_$st.txt; echo "stagger $st: $(tail -1 ${O}_layers_stagger_$st.txt)"
done
for rep in 1 2; do for st in 0 3 5; do
  echo "STAGGER=$st rep $rep $(DFVO_WIN_STAGGER=$st timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg --no-other-legs --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'steady', d['steady_state']['value'])")"
done; done 2>&1 | tee ${O}_stagger_ab.txt
/usr/bin/time -v timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json 2> ${O}_bench_default.err
grep -E "Elapsed|Maximum resident" ${O}_bench_default.err
python -c "
import json; d=json.loads(open('${O}_bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'frac', d['roofline']['frac'])
print('hbm', {k:v for k,v in d['roofline'].get('hbm',{}).items() if k!='other_kernels'})
print('dropin', d['dropin_surface'])
for k,v in (d['other_configs'] or {}).items(): print(k, v)
"
