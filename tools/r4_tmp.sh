#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r4o_taps_offset_table.txt; : > $O
for v in 0 1; do DFVO_TAPS=$v timeout 200 python tools/crc_flow.py 2>&1 | tail -1 | sed "s/^/TAPS=$v /" >> $O; done
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -x -q -k "tap_window or flow" 2>&1 | tail -2 >> $O
rm -f gpurun_out/r4o_layers.csv
DFVO_CONV_PROFILE_CSV=gpurun_out/r4o_layers.csv timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-leg > /dev/null 2>&1
grep "^20," gpurun_out/r4o_layers.csv | awk -F, '{k=$2" "$5"x"$6" cin"$7" cout"$8" k"$9" s"$10" gz"$13; n[k]++; t[k]+=$14} END {for (k in n) printf "  %-46s avg %.1f us x %d\n", k, t[k]/n[k], n[k]/3}' | sort >> $O
for v in 1 0 1 0; do
  echo "== TAPS=$v" >> $O
  DFVO_TAPS=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('frames/s', d['value'], 'steady', d['steady_state']['value'], '| conv family ms', r['conv_family_ms_per_pair'], [ (c['kernel'][:24], c['ms_per_pair']) for c in r['by_config'][:3]])" >> $O
done
cat $O
