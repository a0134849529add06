#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { echo "$@"; env "$@" timeout 300 bash tools/quick.sh and2 2>&1 | tail -1 | cut -c1-170; }
run TQ_DEBUG=0
run TQ_OPT_docsig=0
run TQ_DEBUG=0
