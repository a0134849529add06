#!/bin/bash
# counters of the shared launch ALONE (TQ_AS_PROBE=2: every 2-term query rides in it; with the default 5 % of the
# batch run on and_kernel next to it and the TQ_DEBUG counters of both kernels land in one word)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for keep in 0 1; do
for b in 0 3 7; do
  for d in 0 64 256 32; do
    TQ_AS_PROBE=2 TQ_KEEP_THR=$keep TQ_AS_BOUND=$b TQ_DEBUG=$d timeout 300 python tools/probe_ashare.py 2>&1 | tail -1
  done
done
done > gpurun_out/r5_bounds2.txt 2>&1
cat gpurun_out/r5_bounds2.txt
