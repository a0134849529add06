#!/bin/bash
# quick A/B of one workload on the GPU box: bash tools/quick.sh <workload> [extra bench args]
W=$1; shift
python bench.py --workload $W --no-side --no-cpu-baseline --latency-queries 0 --steps 8 --warmup 2 "$@" 2>&1 | tail -1 | python -c '
import json,sys
j=json.loads(sys.stdin.readline())
print(j["metric"], "qps", j["value"], "ms/step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms_avg"], "frac", j["roofline"]["frac"], "other-mode ms", j["roofline_other_mode"]["kernel_ms_avg"], "scored", j["roofline"].get("docs_scored_per_launch"), "parity", j["pruned_equals_exhaustive"], j["parity_checked_queries"])'
