#!/bin/bash
# HBM-resident data points of the headline kernel:
#   and2 over 8 segments per GPU (1.1 GB of index + side tables, 4x the Infinity Cache), and
#   and2 over a larger vocabulary; each: bench line + rocprofv3 passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --workload and2 --segments 8 --steps 5 --warmup 1 > gpurun_out/bench_and2_s8.log 2>&1
tail -1 gpurun_out/bench_and2_s8.log | cut -c1-1200
KEY_SUFFIX=_s8 bash tools/profile_workload.sh and2 r02_and2_s8 --segments 8 | tail -3
