#!/bin/bash
# round 3, GPU call 2: the shared-union launch — parity first, then A/B against the per-query kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 300 python -m pytest tests -m gpu -x -q -k "or_union or or_matches or mixed_batch or edge_cases") > gpurun_out/c2_tests_a.log 2>&1
grep -E "passed|failed|Aborted|Error" gpurun_out/c2_tests_a.log | tail -3
(time timeout 600 python -m pytest tests -m gpu -x -q) > gpurun_out/c2_tests.log 2>&1
grep -E "passed|failed|Aborted" gpurun_out/c2_tests.log | tail -3
for us in 1 0; do
  echo "or5 TQ_USHARE=$us"; TQ_USHARE=$us timeout 300 bash tools/quick.sh or5 2>&1 | tail -1
  echo "or5 k=10 TQ_USHARE=$us"; TQ_USHARE=$us timeout 300 bash tools/quick.sh or5 --k 10 2>&1 | tail -1
  echo "mixed TQ_USHARE=$us"; TQ_USHARE=$us timeout 300 bash tools/quick.sh mixed 2>&1 | tail -1
done > gpurun_out/c2_ab.log 2>&1
cat gpurun_out/c2_ab.log
