#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
for sk in 1 0; do for d in 32 64 256; do echo -n "skip=$sk TQ_DEBUG=$d "; TQ_US_SKIP_DEAD=$sk TQ_DEBUG=$d bash tools/r6_ab.sh "or5" "base" --steps 2 --warmup 1 | sed -e 's/.*scored/scored/' -e 's/kernels.*//'; done; done
echo "== phases=1 skip=1"; TQ_US_PHASES=1 bash tools/r6_ab.sh "or5 mixed" "base"
echo "== phases=1 skip=0"; TQ_US_PHASES=1 TQ_US_SKIP_DEAD=0 bash tools/r6_ab.sh "or5 mixed" "base"
date
} > gpurun_out/r6_call5.txt 2>&1
