#!/bin/bash
# round 3, GPU call 32: the multi-rank sequence mode of bench.py on real pipelines (2 and 3 ranks sharing device 0, gloo collectives),
# its single-rank equality check, and the RCCL call pattern with world_size 1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 200 python tools/rccl_smoke.py 2>&1 | grep -v amdgpu.ids | tail -3
for n in 2 3; do
  DFVO_BENCH_ONE_DEVICE=1 DFVO_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2> gpurun_out/r3ae_ranks$n.err | tail -1 > gpurun_out/r3ae_ranks$n.json
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r3ae_ranks$n.json').read().strip().splitlines()[-1])
    print('ranks $n (one device):', d['value'], 'n_gpus', d['n_gpus'], d['config']['parallelism'][:60], d['config']['gathered_poses'], d['config']['ranks_seen'], d['sequence_check'])
except Exception as e:
    print('ranks $n failed', e); print(open('gpurun_out/r3ae_ranks$n.err').read()[-2500:])
PY
done
} | tee gpurun_out/r3ae_multirank.txt
