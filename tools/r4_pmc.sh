#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
O=$R/gpurun_out/r4n_pmc_taps_layers.txt; : > $O
L7="N=1 H=352 W=1216 C0=3 COUT=32 K=7"
L17="N=2 H=176 W=608 C0=49 COUT=49 KH=1 KW=7"
LDS71="N=2 H=176 W=608 C0=32 COUT=49 KH=7 KW=1"
LAYER="$L7" bash $R/tools/pmc_layer.sh "7x7 taps" >> $O 2>&1
LAYER="$L7" DFVO_TAPS=0 bash $R/tools/pmc_layer.sh "7x7 generic" >> $O 2>&1
LAYER="$L17" bash $R/tools/pmc_layer.sh "1x7 taps" >> $O 2>&1
LAYER="$L17" DFVO_TAPS=0 bash $R/tools/pmc_layer.sh "1x7 generic" >> $O 2>&1
cat $O
