#!/bin/bash
# round 4: exact-fp32 mode, per-configuration conv time (which launches the fp32 register-ring kernel should take)
mkdir -p gpurun_out
timeout 300 python bench.py --conv-precision fp32 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4d_fp32_by_config.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d_fp32_by_config.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['conv_family_ms_per_pair'])
for c in d['roofline']['by_config']: print(c)
PY
