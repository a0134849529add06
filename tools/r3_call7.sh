#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 600 python -m pytest tests -m gpu -x -q -k "bool or fuzz or deletes or nested or degenerate or global or full_size_boolean or boost") > gpurun_out/c7_tests.log 2>&1
grep -E "passed|failed|Aborted|Error" gpurun_out/c7_tests.log | tail -3
python tools/probe_bool.py 2>&1 | tail -6
timeout 300 bash tools/quick.sh bool 2>&1 | tail -1 | cut -c1-160
TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_uw4.so timeout 300 bash tools/quick.sh bool 2>&1 | tail -1 | cut -c1-160
