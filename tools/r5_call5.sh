#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r5d
DFVO_STREAM_PROBE_VERBOSE=1 timeout 900 python bench.py --steps 20 --warmup 5 > ${O}_bench_default.json 2> ${O}_bench_default.err
grep "stream pool" ${O}_bench_default.err | head -12
python -c "
import json; d=json.loads(open('${O}_bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('hbm', {k:v for k,v in d['roofline'].get('hbm',{}).items() if k not in ('other_kernels','note','kernel')}, len(d['roofline']['hbm'].get('other_kernels') or []))
ds=d['dropin_surface']; print('dropin', ds['value'], ds['stage_ms_per_pair'])
for k,v in (d['other_configs'] or {}).items(): print(k, v.get('value'), v.get('error'))
"
( timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_dropin_gpu.py tests/test_solvers_gpu.py tests/test_pnp_gpu.py -m gpu -q -x 2>&1 | tail -4 ) | tee ${O}_tests.txt
for i in 1 2; do DFVO_STREAM_PROBE_VERBOSE=1 timeout 300 python bench.py --surface mirrors --steps 20 --warmup 3 2>${O}_m$i.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('mirrors', d['value'], d['stage_ms_per_pair'])"; grep "stream pool" ${O}_m$i.err | tail -2; done
