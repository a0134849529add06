#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 600 python -m pytest tests -m gpu -x -q -k "bool or fuzz or deletes or nested or degenerate or global or full_size_boolean or boost") > gpurun_out/c7_tests.log 2>&1
grep -E "passed|failed|Aborted|Error" gpurun_out/c7_tests.log | tail -3
python tools/probe_bool.py 2>&1 | tail -6
echo "B0 off"; TQ_DEBUG=32768 python tools/probe_bool.py 2>&1 | tail -6
echo "cand B"; TQ_DEBUG=128 python tools/probe_bool.py 2>&1 | tail -6
