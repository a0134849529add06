#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
bash tools/r6_ab.sh "or5" "base us_old us_spec us_spec4"
bash tools/r6_ab.sh "mixed" "base us_old us_spec"
bash tools/r6_ab.sh "phrase3 phrase3_adj" "base"
timeout 600 python -m pytest tests/test_gpu_union_sets.py tests/test_gpu_round3.py -x -q 2>&1 | tail -5
date
} > gpurun_out/r6_call1.txt 2>&1
