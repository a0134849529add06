#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 400 python -m pytest tests -m gpu -x -q -k "or_union or or_matches or mixed_batch or edge_cases or fuzz or deletes or sweep or eight or regression or two_full") > gpurun_out/c3_tests_a.log 2>&1
grep -E "passed|failed|Aborted|Error" gpurun_out/c3_tests_a.log | tail -3
run() { echo "$@"; env "$@" timeout 300 bash tools/quick.sh or5 2>&1 | tail -1 | cut -c1-150; }
run TQ_DEBUG=0
run TQ_DEBUG=64
run TQ_DEBUG=128
run TQ_US_PHASE_TASKS=16384
run TQ_US_PHASE_TASKS=4096
run TQ_US_TASK_COST=1024
run TQ_US_TASK_COST=4096
echo mixed; timeout 300 bash tools/quick.sh mixed 2>&1 | tail -1 | cut -c1-150
echo or5 k10; timeout 300 bash tools/quick.sh or5 --k 10 2>&1 | tail -1 | cut -c1-150
bash tools/probe_ushare.sh or5
