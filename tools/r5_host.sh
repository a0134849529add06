#!/bin/bash
# host time of the headline step by phase (TQ_TRACE) + the planner's own phases, then the default quick line
# usage (GPU box): bash tools/r5_host.sh [workload]
W=${1:-and2}
mkdir -p gpurun_out
TQ_TRACE=1 TQ_PLAN_TRACE=1 python bench.py --workload $W --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --no-stream --steps 12 --warmup 3 > gpurun_out/r5_host_$W.json 2> gpurun_out/r5_host_$W.err
grep "\[tq\] plan\|ashare plan" gpurun_out/r5_host_$W.err | tail -24
for i in 1 2; do bash tools/quick2.sh $W --no-pmc-inline --no-stream; done
