#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r4c2 -o kt -- python $R/bench.py --workload and2 --steps 5 --warmup 1 --no-cpu-baseline --latency-queries 0 --no-side > $R/gpurun_out/prof_r4c2.log 2>&1 )
find gpurun_out/prof_r4c2 -name "*kernel_stats.csv" | head -1 | xargs cut -c1-200 | head -14
rm -rf gpurun_out/prof_r4c2/*/*kernel_trace.csv gpurun_out/prof_r4c2/*/*agent_info.csv
for d in 0 32 64 256; do TQ_DEBUG=$d timeout 200 python tools/probe_ashare.py 2>&1 | tail -1; done
TQ_AS_MIN_LEADS=1 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
TQ_AS_MIN_LEADS=16 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
TQ_AS_TASK_PAIRS=256 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
TQ_AS_TASK_PAIRS=1024 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
TQ_AS_TASK_PAIRS=2048 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
TQ_AS_GRID_MUL=16 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
TQ_AS_GRID_MUL=24 timeout 200 python tools/probe_ashare.py 2>&1 | tail -1
