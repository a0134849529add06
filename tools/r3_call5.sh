#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/c5_tests.log 2>&1
grep -E "passed|failed|Aborted" gpurun_out/c5_tests.log | tail -3
(time timeout 900 python bench.py --steps 20 --warmup 3) > gpurun_out/c5_bench.log 2> gpurun_out/c5_bench.err
tail -c 300 gpurun_out/c5_bench.err
bash tools/profile_workload.sh or5 r03_or5 > gpurun_out/c5_prof_or5.log 2>&1; tail -2 gpurun_out/c5_prof_or5.log | cut -c1-200
bash tools/profile_workload.sh mixed r03_mixed > gpurun_out/c5_prof_mixed.log 2>&1; tail -2 gpurun_out/c5_prof_mixed.log | cut -c1-200
