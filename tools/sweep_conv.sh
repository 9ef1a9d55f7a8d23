#!/bin/bash
# M-tile sweep of the implicit-GEMM conv on the main LiteFlowNet / monodepth2 shapes (tuning aid for launch_conv)
cd "$(dirname "$0")/.."
run() { for bm in $BMS; do printf "BM=%-4s " $bm; DFVO_CONV_FORCE_BM=$bm ITERS=10 python tools/bench_conv.py 2>/dev/null | tail -1; done; }
export N=2 K=3 STRIDE=1 C1=0
BMS="128 64 32";  H=96 W=312 C0=128 COUT=128 run; H=48 W=156 C0=128 COUT=128 run; H=24 W=78 C0=256 COUT=128 run
BMS="256 128 64"; H=192 W=624 C0=128 COUT=64 run; H=96 W=312 C0=128 COUT=64 run; H=48 W=156 C0=128 COUT=64 run
BMS="256 128 64"; H=192 W=624 C0=64 COUT=32 run; H=96 W=312 C0=64 COUT=32 run; H=192 W=624 C0=32 COUT=32 run
BMS="256 128 64"; H=384 W=1248 C0=4 COUT=32 K=7 run
N=1; K=3; BMS="256 128 64"; H=96 W=320 C0=64 COUT=64 run; H=48 W=160 C0=64 COUT=64 run
