#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 400 python -m pytest tests -m gpu -x -q -k "or_union or or_matches or mixed_batch or edge_cases or fuzz or deletes or sweep or eight or regression or two_full or kat") > gpurun_out/c3_tests_a.log 2>&1
grep -E "passed|failed|Aborted|Error" gpurun_out/c3_tests_a.log | tail -3
run() { echo "$@"; env "$@" timeout 300 bash tools/quick.sh or5 2>&1 | tail -1 | cut -c1-150;  env "$@" timeout 300 bash tools/quick.sh mixed 2>&1 | tail -1 | cut -c1-150; }
V=$R/tantivy_amd/lib/variants/libtantivy_amd
run TQ_DEBUG=0
run TQ_LIB_PATH=${V}_c2.so
run TQ_LIB_PATH=${V}_c1.so
PHASES_LIST="1 2 4 5 6 9" bash tools/probe_ushare.sh or5
