#!/bin/bash
# round 3 evidence in one call: GPU tests, the default bench line, kernel traces + PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/ev_tests.log 2>&1
grep -E "passed|failed|Aborted" gpurun_out/ev_tests.log | tail -3
(time timeout 900 python bench.py --steps 20 --warmup 3) > gpurun_out/ev_bench.log 2> gpurun_out/ev_bench.err
tail -c 200 gpurun_out/ev_bench.err
bash tools/r3_profiles.sh
