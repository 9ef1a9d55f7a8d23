#!/bin/bash
# End-of-round records on the GPU box (TAG=r6 bash tools/round_records.sh; ~20 GPU-minutes): full -m gpu suite with the
# ANCHOR / FROM-IMAGES accounting lines, smoke, default bench, BASELINE configs 3 / 4 / 5, the CPU-baseline protocol of
# BASELINE.md section 4 (20 pairs after 3 warm-up), the config-3 job on 2 / 3 ranks sharing this box's one device, the flow
# net's error against the float64 anchor per level and per operator.  Outputs under gpurun_out/${TAG}m_*: copy what is
# cited into profiles/.  (tools/profile.sh takes the rocprofv3 records, tools/quick_check.sh is the 3-minute check of a
# kernel change.)
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r6}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "ANCHOR|FROM-IMAGES|F16-MODE|f16x3 range|passed|failed|FAILED|Error" > gpurun_out/${TAG}m_tests.txt
tail -3 gpurun_out/${TAG}m_tests.txt
for prec in f16x3; do
  DFVO_CONV_PRECISION=$prec timeout 300 python tools/flow_error_by_level.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}m_flow_error_by_level_tunnel_$prec.txt
  DFVO_CONV_PRECISION=$prec timeout 300 python tools/flow_op_replay.py --world random 2>&1 | grep -v amdgpu > gpurun_out/${TAG}m_op_replay_random_$prec.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}m_smoke.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}m_bench_default.json
cut -c1-240 gpurun_out/${TAG}m_bench_default.json
timeout 300 python bench.py --sequences kitti-lengths --scale 0.02 2>/dev/null | tail -1 > gpurun_out/${TAG}m_bench_config3_job_1gpu.json
timeout 600 python bench.py --height 960 --width 1280 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}m_bench_config4_1280x960.json
timeout 900 python bench.py --height 1280 --width 1920 --kp-bestn 20000 --e-max-iters 8192 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}m_bench_config5_1920x1280.json
timeout 900 python bench.py --height 1280 --width 1920 --kp-bestn 20000 --e-max-iters 8192 --steps 10 --warmup 3 --no-cpu-baseline --no-exact-leg --conv-precision f16 2>/dev/null | tail -1 > gpurun_out/${TAG}m_bench_config5_f16.json
for s in 0 1; do DFVO_SESSION=$s timeout 300 python bench.py --surface mirrors --steps 30 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${TAG}m_mirrors_session$s.json; done
for f in config3_job_1gpu config4_1280x960 config5_1920x1280 config5_f16 mirrors_session0 mirrors_session1; do python -c "
import json,sys
import glob; d=json.loads(open(glob.glob('gpurun_out/${TAG}m_*$f.json')[0]).read())
print('$f', d['value'], d['ms_per_step'], (d.get('exact_fp32') or {}).get('value'), (d.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))"; done
O=gpurun_out/${TAG}m_multirank_job_one_device.txt; : > $O
for n in 2 3; do
  DFVO_BENCH_ONE_DEVICE=1 DFVO_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510+n)) bench.py --gpus $n --sequences kitti-lengths --scale 0.004 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ranks', d['n_gpus'], 'pairs', d['steps'], 'frames/s (one shared device)', d['value'], '| items per rank', d['config']['items_per_rank'], 'pairs per rank', d['config']['pairs_per_rank'], '|', d['sequence_check'])" >> $O
done
cat $O
timeout 900 python bench.py --cpu-pairs 20 --steps 20 --warmup 5 --no-exact-leg --no-roofline 2>/dev/null | tail -1 > gpurun_out/${TAG}m_bench_cpu_protocol_20pairs.json
python -c "
import json
d=json.loads(open('gpurun_out/${TAG}m_bench_cpu_protocol_20pairs.json').read())
print(d['value'], d['cpu_baseline'])"
