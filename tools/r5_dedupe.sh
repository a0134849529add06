#!/bin/bash
# Round 5: batch-wide dedupe of identical queries (TQ_AS_DEDUPE) x bound on list 1 (TQ_AS_BOUND), parity first
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ashare.py tests/test_gpu_bshare.py tests/test_gpu_submit.py tests/test_gpu_round2.py tests/test_gpu_xunion.py tests/test_gpu_union_sets.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r5_tests2.log 2>&1
cat gpurun_out/r5_tests2.log
{
for dd in 0 1; do for b in 0 3; do
  TQ_AS_DEDUPE=$dd TQ_AS_BOUND=$b bash tools/quick2.sh and2
done; done
for dd in 0 1; do TQ_AS_DEDUPE=$dd bash tools/quick2.sh and2_distinct; done
for dd in 0 1; do TQ_AS_DEDUPE=$dd bash tools/quick2.sh mixed; done
for dd in 0 1; do TQ_AS_DEDUPE=$dd bash tools/quick2.sh bool; done
for d in 0 64 256 32; do TQ_AS_PROBE=2 TQ_DEBUG=$d timeout 300 python tools/probe_ashare.py 2>&1 | tail -1; done
} > gpurun_out/r5_dedupe.txt 2>&1
cat gpurun_out/r5_dedupe.txt
