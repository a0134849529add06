#!/bin/bash
# one line per workload: qps, step, kernel, host (bash tools/quick2.sh <workload> [bench args])
W=$1; shift
python bench.py --workload $W --no-side --no-cpu-baseline --latency-queries 0 --steps ${STEPS:-20} --warmup 3 "$@" 2>&1 | tail -1 | python -c '
import json,sys,os
j=json.loads(sys.stdin.readline()); r=j["roofline"]
print("%-8s threads=%s qps %9.0f ms/step %6.3f kernel %6.3f gpu %s host %6.3f frac %s kernels %s parity %s/%s" % (sys.argv[1], os.environ.get("TQ_PLAN_THREADS","-"), j["value"], j["ms_per_step"], r["kernel_ms_avg"], r.get("gpu_batch_ms"), r["host_plan_ms"], r["frac"], r["kernel"], j["pruned_equals_exhaustive"], j["parity_checked_queries"]))' $W
