#!/bin/bash
# round 3: the default bench command, timed (wall clock of the whole command incl. cpu_baseline and all legs)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
t0=$(date +%s.%N)
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
t1=$(date +%s.%N)
echo "default bench wall seconds: $(echo "$t1 - $t0" | bc)" | tee gpurun_out/r3_bench_default_wall.txt
python - <<'PY' | tee -a gpurun_out/r3_bench_default_wall.txt
import json
d=json.loads(open('gpurun_out/r3_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('default bench:', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'recomputed', d['features_recomputed']['value'])
print('roofline:', r['kernel'][:40], r['achieved'], r['frac'], 'traffic', r.get('traffic'), r.get('traffic_stale'), 'family ms', r['conv_family_ms_per_pair'], 'gflop', r['algorithmic_gflop_per_pair'], r['algorithmic_gflop_per_pair_reference'], 'whole', r['whole_pair_tflops'], r['whole_pair_tflops_reference_work'])
print('cpu_baseline:', d['cpu_baseline'])
PY
