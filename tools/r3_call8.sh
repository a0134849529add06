#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { echo "$@"; env "$@" timeout 300 bash tools/quick.sh phrase3 2>&1 | tail -1 | cut -c1-170; }
V=$R/tantivy_amd/lib/variants/libtantivy_amd
run TQ_DEBUG=0
run TQ_LIB_PATH=${V}_su2w5.so
run TQ_LIB_PATH=${V}_su1w5.so
run TQ_LIB_PATH=${V}_su4w5.so
run TQ_LIB_PATH=${V}_su8w4.so
