#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
date
export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_ust.so
for W in or5 mixed; do
for ph in 1 2 3 4 5 6 7 8 9 10 11 12 13; do
  echo -n "$W region $ph: "
  TQ_DEBUG=$((ph<<16)) python bench.py --workload $W --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --no-stream --check-queries 8 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c '
import json,sys
j=json.loads(sys.stdin.readline()); print(j["roofline"]["docs_scored_per_launch"], "kernel_ms", j["roofline"]["kernel_ms_avg"])'
done; done
date
} > gpurun_out/r6_call3.txt 2>&1
