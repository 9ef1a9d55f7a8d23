#!/bin/bash
# round 3, final verification: full -m gpu suite + smoke() on HEAD, the default bench line (timed), host-fed frames, the
# drop-in surface, then the rocprofv3 kernel-trace / counter passes of the default command (TAG=r3 -> profiles/r3_*)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r3_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r3_final_tests.txt
TAG=r3 bash tools/r3_profile.sh 2>&1 | tail -12
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/r3_pmc_bench.json profiles/r3_pmc_bench.json   # (the default bench below reads the traffic of THIS build)
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
t1=$(date +%s)
echo "default bench wall seconds: $((t1 - t0))" | tee gpurun_out/r3_bench_default_wall.txt
python - <<'PY' | tee -a gpurun_out/r3_bench_default_wall.txt
import json
d=json.loads(open('gpurun_out/r3_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('default bench:', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'recomputed', d['features_recomputed']['value'], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])
print('roofline:', r['kernel'][:40], r['achieved'], r['frac'], 'traffic', r.get('traffic'), r.get('traffic_stale'), 'family ms', r['conv_family_ms_per_pair'], 'gflop', r['algorithmic_gflop_per_pair'], r['algorithmic_gflop_per_pair_reference'], 'whole', r['whole_pair_tflops'], r['whole_pair_tflops_reference_work'])
print('cpu_baseline:', d['cpu_baseline'])
PY
timeout 300 python bench.py --frames host --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null > gpurun_out/r3_frames_host.json
python -c "
import json
d=json.loads(open('gpurun_out/r3_frames_host.json').read().strip().splitlines()[-1]); print('frames host', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['frames'])" | tee -a gpurun_out/r3_final_tests.txt
timeout 300 python bench.py --surface mirrors --conv-precision f16x3 --steps 20 --warmup 3 > gpurun_out/r3_mirrors_f16x3.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r3_mirrors_f16x3.json').read().strip().splitlines()[-1]); print('mirrors f16x3', d['value'], d['ms_per_step'], d['stage_ms_per_pair'])" | tee -a gpurun_out/r3_final_tests.txt
