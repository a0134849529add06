#!/bin/bash
# the round's last GPU call: the default bench line, then the unpruned or5 / mixed profiles (doc-major union launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(timeout 85 python bench.py --steps 20 --warmup 3) > gpurun_out/ev_bench.log 2> gpurun_out/ev_bench.err
tail -c 300 gpurun_out/ev_bench.log
timeout 62 bash tools/profile_workload.sh or5 r03_or5_exhaustive --exhaustive > gpurun_out/prof_or5x.log 2>&1
timeout 75 bash tools/profile_workload.sh mixed r03_mixed_exhaustive --exhaustive > gpurun_out/prof_mixedx.log 2>&1
rm -f gpurun_out/prof_*/*/*agent_info.csv
du -sh gpurun_out
