#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_tree.py tests/test_gpu_bshare.py tests/test_gpu_round5.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30
date
} > gpurun_out/r6_call10.txt 2>&1
