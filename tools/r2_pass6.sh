#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -q -m gpu -x -rs > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -5 $O/gpu_suite.log; grep -a "HIP    (\|oracle        :\|per-pair\|modes \[" $O/gpu_suite.log | head
python -m pytest tests/test_trajectory_gpu.py tests/test_dropin_gpu.py -q -m gpu -s -k "trajectory or main_loop" 2>&1 | grep -a "HIP  \|oracle   \|per-pair\|modes\|passed\|failed" | head -12
python bench.py --steps 30 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --solver-inputs synthetic > $O/bench_syn.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --height 384 --width 1248 > $O/bench_mux.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --conv-precision fp32 > $O/bench_fp32.json 2>/dev/null
python - <<PY
import json
for f in ("bench_default","bench_syn","bench_mux","bench_fp32"):
    d=json.load(open("$O/"+f+".json")); r=d["roofline"]
    print(f, d["value"], "fps", d["ms_per_step"], "ms | E/PnP", d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"], "| fam", r["conv_family_achieved"], "TF/s", r["conv_family_ms_per_pair"], "ms | GF", r["algorithmic_gflop_per_pair"], "|", r["kernel"][:30], r["achieved"], r["frac"])
PY
