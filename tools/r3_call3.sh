#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { echo "$@"; env "$@" timeout 300 bash tools/quick.sh or5 2>&1 | tail -1 | cut -c20-150; env "$@" timeout 300 bash tools/quick.sh mixed 2>&1 | tail -1 | cut -c20-150; }
for pt in 3072 4096 5120 6144 7168; do run TQ_US_PHASE_TASKS=$pt; done
