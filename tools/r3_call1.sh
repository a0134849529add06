#!/bin/bash
# round 3, GPU call 1: tests, the full bench line, exhaustive unions through the candidate kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/c1_tests.log 2>&1
tail -5 gpurun_out/c1_tests.log
(time timeout 900 python bench.py --steps 20 --warmup 3) > gpurun_out/c1_bench.log 2> gpurun_out/c1_bench.err
tail -c 600 gpurun_out/c1_bench.err; tail -c 300 gpurun_out/c1_bench.log
for ow in -1 0; do
  echo "or5 or_windows=$ow"; TQ_OPT_or_windows=$ow bash tools/quick.sh or5 2>&1 | tail -1
  echo "mixed or_windows=$ow"; TQ_OPT_or_windows=$ow bash tools/quick.sh mixed 2>&1 | tail -1
done > gpurun_out/c1_orw.log 2>&1
cat gpurun_out/c1_orw.log
