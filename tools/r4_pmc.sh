#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
O=$R/gpurun_out/r4q_pmc_window_layers.txt; : > $O
LAYER="N=2 H=176 W=608 C0=128 COUT=128 K=3" bash $R/tools/pmc_layer.sh "L2 128->128 v2" >> $O 2>&1
LAYER="N=2 H=176 W=608 C0=128 COUT=64 K=3" bash $R/tools/pmc_layer.sh "L2 128->64 v2" >> $O 2>&1
LAYER="N=2 H=176 W=608 C0=64 COUT=32 K=3" bash $R/tools/pmc_layer.sh "L2 64->32 v2" >> $O 2>&1
cat $O
