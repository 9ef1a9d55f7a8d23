#!/bin/bash
# round 4, GPU call 1: full -m gpu suite on the lane-parallel Durand-Kerner build; A/B of the polynomial stage in the bench
# (DFVO_E_POLY_LANES 1 = one root per lane + stage 3 fused, 0 = round 3's one lane per hypothesis); config-3 job mode
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r4a_tests.txt
for v in 1 0 1 0; do
  echo "== DFVO_E_POLY_LANES=$v" >> gpurun_out/r4a_poly_ab.txt
  DFVO_E_POLY_LANES=$v DFVO_TRACK_TRACE=1 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-exact-leg \
    2>gpurun_out/r4a_err_$v.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('frames/s', d['value'], 'ms', d['ms_per_step'], 'steady', d['steady_state']['value'], 'E', d['config']['tracked_by_E'], 'PnP', d['config']['tracked_by_PnP'])" >> gpurun_out/r4a_poly_ab.txt
  grep "track device ms" gpurun_out/r4a_err_$v.txt | tail -2 >> gpurun_out/r4a_poly_ab.txt
done
timeout 300 python bench.py --sequences kitti-lengths --scale 0.02 > gpurun_out/r4a_job_config3.json 2> gpurun_out/r4a_job_err.txt
tail -3 gpurun_out/r4a_job_err.txt
cat gpurun_out/r4a_tests.txt | tail -8
cat gpurun_out/r4a_poly_ab.txt
cat gpurun_out/r4a_job_config3.json | cut -c1-600
