#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/c3_tests.log 2>&1
grep -E "passed|failed|Aborted|Error" gpurun_out/c3_tests.log | tail -3
for w in and2 or5 mixed bool phrase3; do timeout 300 bash tools/quick.sh $w 2>&1 | tail -1 | cut -c1-170; done
