#!/bin/bash
# Round 5: task size of the shared intersections after the batch-wide dedupe (fewer leads -> fewer, longer tasks)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for tp in 512 256 128 64; do for b in 3 7; do
  TQ_AS_TASK_PAIRS=$tp TQ_AS_BOUND=$b bash tools/quick2.sh and2
done; done
for tp in 256 128; do TQ_AS_TASK_PAIRS=$tp TQ_AS_WARM_PERMILLE=5 bash tools/quick2.sh and2; done
for tp in 512 256 128 64; do TQ_AS_TASK_PAIRS=$tp bash tools/quick2.sh and2_distinct; done
for tp in 128 ; do for d in 0 64 256 32; do TQ_AS_TASK_PAIRS=$tp TQ_AS_PROBE=2 TQ_DEBUG=$d timeout 300 python tools/probe_ashare.py 2>&1 | tail -1; done; done
} > gpurun_out/r5_tasks.txt 2>&1
cat gpurun_out/r5_tasks.txt
