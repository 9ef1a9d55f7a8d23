#!/bin/bash
TAG=${TAG:-r6}
# last records of a round on the final build (TAG=r6 bash tools/final_records.sh; ~2 GPU-minutes): the driver's command (all legs), the exact-fp32 roofline table, three class-surface runs
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}z_bench_driver_protocol.json
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}z_bench_driver_protocol.json').read())
print('driver protocol', d['value'], 'exact', d['exact_fp32']['value'], 'frames_host', d['frames_host']['value'], 'dropin', d['dropin_surface']['value'], 'frac', d['roofline']['frac'], 'hbm', d['roofline']['hbm']['frac'], {k: v.get('value') for k, v in d['other_configs'].items()})"
timeout 600 python bench.py --conv-precision fp32 --steps 30 --warmup 5 --no-cpu-baseline --no-other-legs --no-exact-leg 2>/dev/null | tail -1 > gpurun_out/${TAG}z_bench_fp32_roofline.json
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}z_bench_fp32_roofline.json').read()); r=d['roofline']
print('fp32', d['value'], 'steady', d['steady_state']['value'], 'family TF/s', r['conv_family_achieved'], 'family ms', r['conv_family_ms_per_pair'])
for c in r['by_config']: print('  ', c['kernel'][:70], c['ms_per_pair'], c['launches_per_pair'], c['gflop_per_pair'], c['tflops'])"
for i in 1 2 3; do
timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}z_mirrors_run$i.json
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}z_mirrors_run$i.json').read()); print('mirrors', d['value'], d['stage_ms_per_pair'])"
done
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-other-legs 2>/dev/null | tail -1 > gpurun_out/${TAG}z_bench_200steps.json
python -c "
import json; d=json.loads(open('gpurun_out/${TAG}z_bench_200steps.json').read()); print('200 steps', d['value'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'])"
timeout 600 python -m pytest tests/test_resize_gpu.py tests/test_dropin_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
