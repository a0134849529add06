#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/c9_tests.log 2>&1
grep -E "passed|failed|Aborted" gpurun_out/c9_tests.log | tail -3
(time timeout 900 python bench.py --steps 20 --warmup 3) > gpurun_out/c9_bench.log 2> gpurun_out/c9_bench.err
tail -c 200 gpurun_out/c9_bench.err
