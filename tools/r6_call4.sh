#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
echo "== new (group-major + skip + wide F)"; bash tools/r6_ab.sh "or5 mixed" "base"
echo "== old order"; TQ_US_BLOCK_MAJOR=1 bash tools/r6_ab.sh "or5 mixed" "base"
echo "== group-major, no skip"; TQ_US_SKIP_DEAD=0 bash tools/r6_ab.sh "or5 mixed" "base"
echo "== new order, old F"; bash tools/r6_ab.sh "or5 mixed" "us_fold"
echo "== old order, old F"; TQ_US_BLOCK_MAJOR=1 bash tools/r6_ab.sh "or5 mixed" "us_fold"
timeout 600 python -m pytest tests/test_gpu_union_sets.py tests/test_gpu_round3.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
date
} > gpurun_out/r6_call4.txt 2>&1
