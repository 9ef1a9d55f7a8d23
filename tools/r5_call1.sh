#!/bin/bash
# round 5, GPU call 1: correctness of the f16 mode and of the single-accumulator window kernel (DFVO_WIN=a), then same-box A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r5a
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -s -k "f16_mode or dynamic_range" 2>&1 | grep -E "f16x3 range|conv f16|passed|failed|FAILED|Error|assert" | tail -150 ) > ${O}_ops_f16.txt
tail -3 ${O}_ops_f16.txt
( DFVO_WIN=a timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "test_conv or f16x3_window" 2>&1 | tail -15 ) > ${O}_ops_win_a.txt
tail -3 ${O}_ops_win_a.txt
( DFVO_WIN=a timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q -x -s -k "f16x3" 2>&1 | grep -E "ANCHOR|passed|failed|FAILED|Error|assert" | tail -40 ) > ${O}_nets_win_a.txt
tail -3 ${O}_nets_win_a.txt
( timeout 900 python -m pytest tests/test_f16_mode_gpu.py "tests/test_e2e_gpu.py::test_pipeline_config5_settings" -m gpu -q -s 2>&1 | grep -E "F16-MODE|config 5|passed|failed|FAILED|Error|assert" | tail -40 ) > ${O}_f16_mode.txt
tail -5 ${O}_f16_mode.txt
for v in 2 a; do
  DFVO_WIN=$v timeout 200 python tools/bench_window_layers.py 2>/dev/null | grep -v amdgpu > ${O}_layers_win_$v.txt
  tail -1 ${O}_layers_win_$v.txt
done
PREC=f16 timeout 200 python tools/bench_window_layers.py 2>/dev/null | grep -v amdgpu > ${O}_layers_f16.txt; tail -1 ${O}_layers_f16.txt
for rep in 1 2; do for v in 2 a; do
  DFVO_WIN=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg 2>/dev/null | tail -1 > ${O}_bench_win_${v}_$rep.json
  python - <<PY
import json
d=json.loads(open("${O}_bench_win_${v}_$rep.json").read())
r=d['roofline']
print('DFVO_WIN=$v rep $rep frames/s', d['value'], 'steady', d['steady_state']['value'], '| frac', r['frac'], 'conv ms', r['conv_family_ms_per_pair'], [(c['kernel'][:14], c['ms_per_pair']) for c in r['by_config'][:3]], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])
PY
done; done 2>&1 | tee ${O}_bench_ab.txt
timeout 300 python bench.py --conv-precision f16 --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg 2>/dev/null | tail -1 > ${O}_bench_f16.json
python -c "
import json
d=json.loads(open('${O}_bench_f16.json').read()); r=d['roofline']
print('f16 frames/s', d['value'], 'steady', d['steady_state']['value'], 'frac', r['frac'], 'conv ms', r['conv_family_ms_per_pair'], d['config']['tracked_by_E'], d['dtype'][:30])"
