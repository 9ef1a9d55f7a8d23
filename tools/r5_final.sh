#!/bin/bash
# round 5, final records after the solver-side work (Jacobi, pose ahead, per-keypoint depth): the driver's command, the class
# surface in the three precisions and without the session, then the rocprofv3 records of tools/profile.sh
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; T=r5z
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${T}_bench_driver_protocol.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5z_bench_driver_protocol.json").read())
print("driver protocol:", d["value"], d["ms_per_step"], "steady", (d.get("steady_state") or {}).get("value"), "exact", (d.get("exact_fp32") or {}).get("value"),
      "frac", d["roofline"]["frac"], "hbm", (d["roofline"].get("hbm") or {}).get("frac"), "cpu", d["cpu_baseline"]["value"])
for k, v in (d.get("other_configs") or {}).items():
    print("   leg", k, v if not isinstance(v, dict) else {a: v.get(a) for a in ("value", "ms_per_step", "dtype", "error") if a in v})
print("   dropin_surface", d.get("dropin_surface"))
PY
for spec in "1 f16x3" "0 f16x3" "1 fp32" "1 f16"; do set -- $spec
  DFVO_SESSION=$1 timeout 300 python bench.py --surface mirrors --steps 30 --warmup 3 --conv-precision $2 2>/dev/null | tail -1 > gpurun_out/${T}_mirrors_session$1_$2.json
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_mirrors_session$1_$2.json').read()); print('mirrors session=$1 $2:', d['value'], d['stage_ms_per_pair'], d['session'])"
done
DFVO_SESSION_POSE_AHEAD=0 timeout 300 python bench.py --surface mirrors --steps 30 --warmup 3 2>/dev/null | tail -1 > gpurun_out/${T}_mirrors_no_pose_ahead.json
python -c "
import json; d=json.loads(open('gpurun_out/${T}_mirrors_no_pose_ahead.json').read()); print('mirrors, pose ahead off:', d['value'], d['stage_ms_per_pair'])"
timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-exact-leg --no-other-legs 2>/dev/null | tail -1 > gpurun_out/${T}_bench_200steps.json
python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_200steps.json').read()); print('200 steps:', d['value'], 'steady', (d.get('steady_state') or {}).get('value'), 'frac', d['roofline']['frac'])"
TAG=$T bash tools/profile.sh > gpurun_out/${T}_profile.log 2>&1; tail -12 gpurun_out/${T}_profile.log | cut -c1-200
