#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 300 python bench.py --surface mirrors --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r6a_mirrors_base.json
python -c "
import json; d=json.loads(open('gpurun_out/r6a_mirrors_base.json').read()); print(d['value'], d['stage_ms_per_pair'])"
TAG=r6a bash tools/timeline_detail.sh
