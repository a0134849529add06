#!/bin/bash
# Round 5: A/B of the bound on list 1 of the shared intersections (TQ_AS_BOUND bits: 1 block pre-filter, 2 gather
# cut, 4 per-doc test; 0 = round 4) with the work counters of the headline batch, and the floor with the previous
# batch's thresholds kept (TQ_KEEP_THR=1).  Everything inside ONE call (boxes differ by +-5 %).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ashare.py tests/test_gpu_bshare.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r5_tests1.log 2>&1
cat gpurun_out/r5_tests1.log
for b in 0 1 3 7 5 4; do
  for d in 0 64 256 32; do
    TQ_AS_BOUND=$b TQ_DEBUG=$d timeout 300 python tools/probe_ashare.py 2>&1 | tail -1
  done
done > gpurun_out/r5_bounds.txt 2>&1
for b in 0 7; do
  for d in 0 64 256; do
    TQ_KEEP_THR=1 TQ_AS_BOUND=$b TQ_DEBUG=$d timeout 300 python tools/probe_ashare.py 2>&1 | tail -1
  done
done >> gpurun_out/r5_bounds.txt 2>&1
cat gpurun_out/r5_bounds.txt
