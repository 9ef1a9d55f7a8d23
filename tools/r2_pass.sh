#!/bin/bash
# one GPU-box pass: e2e tests, bench in both solver-input modes (args: "suite" to also run the full -m gpu suite)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -s > $O/e2e.log 2>&1; echo "e2e rc $?" | tee -a $O/e2e.log
if [ "$1" == "suite" ]; then
python -m pytest tests -q -m gpu -x --deselect tests/test_e2e_gpu.py > $O/gpu_suite.log 2>&1; echo "suite rc $?" | tee -a $O/gpu_suite.log
fi
python bench.py --steps 30 --warmup 5 > $O/bench_nets.json 2> $O/bench_nets.err; echo "bench nets rc $?"
python bench.py --steps 30 --warmup 5 --solver-inputs synthetic --no-cpu-baseline > $O/bench_syn.json 2> $O/bench_syn.err; echo "bench syn rc $?"
python bench.py --steps 30 --warmup 5 --height 384 --width 1248 --no-cpu-baseline > $O/bench_nets_mux.json 2> $O/bench_nets_mux.err; echo "bench nets mux rc $?"
grep -a "passed\|failed" $O/e2e.log | tail -3
for f in bench_nets bench_syn bench_nets_mux; do python - <<PY
import json
d=json.load(open("$GRAFT_REPO_ROOT/$O/$f.json"))
print("$f", d["value"], "fps", d["ms_per_step"], "ms; E/PnP", d["config"]["tracked_by_E"], d["config"]["tracked_by_PnP"], "| conv fam", d["roofline"]["conv_family_achieved"], "TF/s", d["roofline"]["conv_family_ms_per_pair"], "ms")
PY
done
