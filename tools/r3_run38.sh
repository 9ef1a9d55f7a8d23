#!/bin/bash
# round 3: full -m gpu suite + smoke() on HEAD once more (after the ordering / guard / flag changes), default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r3_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r3_final_tests.txt
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
python - <<'PY' | tee -a gpurun_out/r3_final_tests.txt
import json
d=json.loads(open('gpurun_out/r3_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('default bench:', d['value'], d['ms_per_step'], 'steady', d['steady_state']['value'], 'exact', d['exact_fp32']['value'], 'recomputed', d['features_recomputed']['value'], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])
print('roofline:', r['achieved'], r['frac'], 'traffic', r.get('traffic'), r.get('traffic_stale'), 'family ms', r['conv_family_ms_per_pair'])
PY
