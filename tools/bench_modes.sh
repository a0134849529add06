#!/bin/bash
# side workloads, one JSON line each (kernel ms / QPS only): tools/bench_modes.sh [extra bench args]
for w in "or5 --queries 1000" "mixed --queries 10000" "bool --queries 2000" "phrase3 --queries 1000"; do
  python bench.py --workload $w --no-cpu-baseline --latency-queries 0 "$@" 2>&1 | tail -1 | python -c '
import json,sys
j=json.loads(sys.stdin.readline())
print(j["metric"], "qps", j["value"], "ms/step", j["ms_per_step"], "kernel_ms", j["roofline"]["kernel_ms_avg"], "frac", j["roofline"]["frac"], "other", j["roofline_other_mode"]["kernel_ms_avg"], "scored", j["roofline"].get("docs_scored_per_launch"))'
done
