cd $GRAFT_REPO_ROOT
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -4
(N=2 H=192 W=624 C0=128 COUT=128 K=3 STRIDE=1 ITERS=6000 python tools/bench_conv.py > /tmp/bc.log 2>&1) &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
tail -1 /tmp/bc.log
