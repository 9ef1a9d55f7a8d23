#!/bin/bash
# round 4: persistent window kernel with reserved CUs (DFVO_F16S_RUN=p<k>: one workgroup per CU on 256 - k CUs) vs the
# second skeleton, same box, alternating
mkdir -p gpurun_out
O=gpurun_out/r4h_persistent_reserve_ab.txt; : > $O
for r in 0 p16 p32 p 0 p16 p32; do
  echo "== DFVO_F16S_RUN='$r'" >> $O
  DFVO_F16S_RUN=$r timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('frames/s', d['value'], 'steady', d['steady_state']['value'], '| frac', r['frac'], 'avg us', r['avg_launch_us'], 'conv family ms', r['conv_family_ms_per_pair'])" >> $O
done
cat $O
