#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -q -m gpu -x --deselect tests/test_tracker_gpu.py --deselect tests/test_solvers_gpu.py --deselect tests/test_pnp_gpu.py --deselect tests/test_dropin_gpu.py --deselect tests/test_lanczos_gpu.py > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -4 $O/gpu_suite.log
rm -f /tmp/conv.csv
for m in f16x3; do
DFVO_CONV_PROFILE_CSV=/tmp/conv.csv python bench.py --steps 30 --warmup 5 --no-cpu-baseline --solver-inputs synthetic --conv-precision $m > $O/b_syn_$m.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --conv-precision $m > $O/b_nets_$m.json 2>/dev/null
python - <<PY
import json
for f in ("b_syn_$m","b_nets_$m"):
    d=json.load(open("$O/"+f+".json")); r=d["roofline"]
    print(f, d["value"], "fps", d["ms_per_step"], "ms | E/PnP", d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"], "| fam", r["conv_family_achieved"], "TF/s", r["conv_family_ms_per_pair"], "ms")
d=json.load(open("$O/b_syn_$m.json"))
for c in d["roofline"]["by_config"]: print("   %-60s %6.3f ms %3d launches %6.1f GF %6.1f TF/s"%(c["kernel"][:60],c["ms_per_pair"],c["launches_per_pair"],c["gflop_per_pair"],c["tflops"]))
PY
done
python tools/conv_csv_agg.py /tmp/conv.csv > $O/conv_layers_f16x3.txt; head -60 $O/conv_layers_f16x3.txt
