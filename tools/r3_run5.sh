#!/bin/bash
# round 3, GPU call 5: compile-time ablations of the one-wave-per-SIMD skeleton (128 couts x 6 rows), level-2 128->128
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
sed -i 's/^LAYERS = \[/LAYERS_ALL = [/' tools/bench_f16s_v2.py
sed -i 's/^g = torch.Generator/LAYERS = [l for l in LAYERS_ALL if l[0] in os.environ.get("ONLY", "L2 128->128").split(",")]\ng = torch.Generator/' tools/bench_f16s_v2.py
for sk in 0 1 2 4 3 5 6 7; do
  echo -n "SKIP=$sk (1 weights, 2 LDS reads, 4 window): "; DFVO_F16S_V2=1 DFVO_F16S2_SKIP=$sk timeout 200 python tools/bench_f16s_v2.py 2>&1 | grep -v amdgpu.ids | head -1
done | tee gpurun_out/r3e_skip_ablation.txt
