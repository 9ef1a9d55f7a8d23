#!/bin/bash
# round 3, GPU call 7: full -m gpu suite on the current build, then stage view, drop-in surface and host-frame timings,
# A/B of the two window skeletons inside the pipeline (repeat), rocprofv3 kernel trace of the default command
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r3g_tests.log
for v in 1 0 1 0; do
  DFVO_F16S_V2=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2=$v', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'])"
done | tee gpurun_out/r3g_v2_ab.txt
DFVO_TRACK_TRACE=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg 2> gpurun_out/r3g_track_trace.txt > /dev/null
grep -E "track (device|host) ms" gpurun_out/r3g_track_trace.txt | tail -8
STEPS=30 timeout 300 python tools/bench_stages.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r3g_stages.txt
for prec in fp32 f16x3; do
  timeout 300 python bench.py --surface mirrors --conv-precision $prec --steps 20 --warmup 3 > gpurun_out/r3g_mirrors_$prec.json 2> gpurun_out/r3g_mirrors_$prec.err
  python -c "
import json
try:
    d=json.loads(open('gpurun_out/r3g_mirrors_$prec.json').read().strip().splitlines()[-1]); print('mirrors $prec', d['value'], d['ms_per_step'], d['stage_ms_per_pair'], d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])
except Exception as e: print('mirrors $prec failed', e, open('gpurun_out/r3g_mirrors_$prec.err').read()[-1500:])"
done
timeout 300 python bench.py --frames host --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg > gpurun_out/r3g_frames_host.json 2> gpurun_out/r3g_frames_host.err
python -c "
import json
try:
    d=json.loads(open('gpurun_out/r3g_frames_host.json').read().strip().splitlines()[-1]); print('frames host', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], d['config']['frames'])
except Exception as e: print('frames host failed', e, open('gpurun_out/r3g_frames_host.err').read()[-1500:])"
cd /tmp && rm -rf /tmp/prof_r3g && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3g -o r3g -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-exact-leg > /tmp/prof_out.txt 2> /tmp/prof_err.txt
cd "$GRAFT_REPO_ROOT"
find /tmp/prof_r3g -type f | head -20
f=$(find /tmp/prof_r3g -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r3g_rocprofv3_kernel_stats.csv; head -30 gpurun_out/r3g_rocprofv3_kernel_stats.csv | cut -c1-150; else tail -5 /tmp/prof_err.txt; fi
