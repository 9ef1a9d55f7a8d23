cd /root/repo
for sh in "N=2 H=192 W=624 C0=128 COUT=128" "N=2 H=192 W=624 C0=128 C1=4 COUT=128" "N=2 H=192 W=624 C0=128 COUT=64" "N=2 H=192 W=624 C0=64 COUT=32" "N=2 H=192 W=624 C0=32 COUT=32" "N=2 H=96 W=312 C0=128 COUT=128" "N=2 H=96 W=312 C0=128 COUT=64" "N=2 H=96 W=312 C0=64 COUT=32" "N=1 H=96 W=320 C0=64 COUT=64"; do
  for m in 0 1; do printf "win=%d " $m; env $sh K=3 STRIDE=1 ITERS=10 DFVO_CONV_WINDOW=$m python tools/bench_conv.py 2>/dev/null | tail -1; done; done
