#!/bin/bash
# round 5: barrier-free window loop (four LDS buffers + slack counter, DFVO_WIN_NB=4) against the barrier form
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out/r5g
for nb in 2 4; do DFVO_WIN_NB=$nb timeout 200 python tools/bench_window_layers.py 2>/dev/null | grep -v amdgpu > ${O}_layers_nb$nb.txt; echo "NB=$nb: $(tail -1 ${O}_layers_nb$nb.txt)"; done
paste <(cut -c1-60 ${O}_layers_nb2.txt) <(cut -c18-60 ${O}_layers_nb4.txt) | head -12
( DFVO_WIN_NB=4 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "(test_conv and f16x3) or f16x3_window or dynamic_range" 2>&1 | tail -3 ) | tee ${O}_ops_nb4.txt
( DFVO_WIN_NB=4 timeout 600 python -m pytest tests/test_nets_gpu.py -m gpu -q -x -k "f16x3 and (vs_oracle or exact_function)" 2>&1 | tail -3 ) | tee ${O}_nets_nb4.txt
for rep in 1 2 3; do for nb in 2 4; do
  echo "NB=$nb rep $rep $(DFVO_WIN_NB=$nb timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg --no-other-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'steady', d['steady_state']['value'], 'frac', r['frac'], 'win ms', r['by_config'][0]['ms_per_pair'], 'E/PnP', d['config']['tracked_by_E'], d['config']['tracked_by_PnP'])")"
done; done 2>&1 | tee ${O}_bench_ab.txt
