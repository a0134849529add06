#!/bin/bash
# host planning under contention: the headline and the mixed batch on the GPU
# while 7 more processes plan batches in a loop on the same granted CPUs (as 8 ranks of a node would).
#   HERE first: bash tools/planbench/build.sh && cp /tmp/plan_bench tools/planbench/bin/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
echo "granted CPUs: $(cat /sys/fs/cgroup/cpu.max)"
for load in 0 7; do
  pids=""
  for i in $(seq 1 $load); do tools/planbench/bin/plan_bench 10000 100000 ashare > /dev/null 2>&1 & pids="$pids $!"; done
  for w in and2 mixed; do echo -n "background planners $load: "; STEPS=30 bash tools/quick2.sh $w; done
  for p in $pids; do kill $p 2>/dev/null; done
  wait 2>/dev/null
done
