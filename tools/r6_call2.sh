#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
echo "== keep thr"
TQ_KEEP_THR=1 bash tools/r6_ab.sh "or5 mixed" "us_old"
echo "== counters (us_old): 32 pairs, 64 candidates, 128 block searches, 256 blocks decoded, 512 stage-C steps"
for d in 32 64 128 256 512; do echo -n "TQ_DEBUG=$d "; TQ_DEBUG=$d bash tools/r6_ab.sh "or5" "us_old" --steps 2 --warmup 1 | sed -e 's/.*scored/scored/' -e 's/kernels.*//'; done
for d in 32 64 128 256 512; do echo -n "mixed TQ_DEBUG=$d "; TQ_DEBUG=$d TQ_ASHARE=0 bash tools/r6_ab.sh "mixed" "us_old" --steps 2 --warmup 1 | sed -e 's/.*scored/scored/' -e 's/kernels.*//'; done
echo "== counters (base = deferral): 1024 parked"
for d in 64 1024; do echo -n "TQ_DEBUG=$d "; TQ_DEBUG=$d bash tools/r6_ab.sh "or5" "base" --steps 2 --warmup 1 | sed -e 's/.*scored/scored/' -e 's/kernels.*//'; done
date
} > gpurun_out/r6_call2.txt 2>&1
