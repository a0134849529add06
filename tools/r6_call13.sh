#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
TQ_TRACE=1 python bench.py --workload and2 --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --steps 3 --warmup 1 --stream-vocabs 65536 2>&1 | grep -i "tq\]\|trace\|plan" | tail -40
date
} > gpurun_out/r6_call13.txt 2>&1
