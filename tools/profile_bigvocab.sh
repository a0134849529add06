#!/bin/bash
# the HBM-resident data point: and2 over a 4096-term (or $1-term) vocabulary — few repeated queries,
# most lists without a bitmap; bench line + rocprofv3 passes (tools/profile_workload.sh)
T=${1:-4096}
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --workload and2 --terms $T --no-cpu-baseline --latency-queries 0 --steps 10 --warmup 2 > gpurun_out/bench_and2_t$T.log 2>&1
tail -1 gpurun_out/bench_and2_t$T.log | cut -c1-1500
bash tools/profile_workload.sh and2 r02_and2_t$T --terms $T
