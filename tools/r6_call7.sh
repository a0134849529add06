#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
bash tools/r6_ab.sh "phrase3 phrase3_adj" "base ph_old ph_w4"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "phrase" 2>&1 | tail -3
date
} > gpurun_out/r6_call7.txt 2>&1
