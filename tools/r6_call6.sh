#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
for r in 64 128 256 32; do echo "== TQ_PH_SWEEP_RATIO=$r"; TQ_PH_SWEEP_RATIO=$r bash tools/r6_ab.sh "phrase3" "base"; done
date
} > gpurun_out/r6_call6.txt 2>&1
