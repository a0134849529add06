#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
echo "== dead flags on"; bash tools/r6_ab.sh "or5 mixed" "base"
echo "== dead flags off"; TQ_US_DEAD_FLAGS=0 bash tools/r6_ab.sh "or5 mixed" "base"
timeout 600 python -m pytest tests/test_gpu_union_sets.py tests/test_gpu_round3.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
date
} > gpurun_out/r6_call15.txt 2>&1
