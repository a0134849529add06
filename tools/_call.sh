#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
bash tools/ab_bench.sh "phrase3 phrase3_adj" "base"
echo "== TQ_DOCCLS=0"; TQ_DOCCLS=0 bash tools/ab_bench.sh "phrase3 phrase3_adj" "base"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_tree.py -x -q -k "phrase or Phrase or position or compat or tree" 2>&1 | tail -3
date
} > gpurun_out/r6_call16.txt 2>&1
