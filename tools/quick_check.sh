#!/bin/bash
# quick GPU check of a kernel change: CRC of the net outputs, op / net parity tests, default bench summary
mkdir -p gpurun_out
timeout 200 python tools/crc_flow.py 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-exact-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('frames/s', d['value'], 'steady', d['steady_state']['value'], '| frac', r['frac'], 'conv family ms', r['conv_family_ms_per_pair'], [(c['kernel'][:22], c['ms_per_pair']) for c in r['by_config'][:4]])"
done
