#!/bin/bash
# A/B of library variants on bench workloads inside ONE gpurun call: bash tools/ab_bench.sh "<workloads>" "<variants|base>" [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WLS=$1; VARS=$2; shift; shift
for w in $WLS; do
  for v in $VARS; do
    if [ "$v" = base ]; then unset TQ_LIB_PATH; else export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_$v.so; fi
    echo -n "$w $v: "
    timeout 300 python bench.py --workload $w --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --no-stream --steps 10 --warmup 2 --check-queries 64 "$@" 2>gpurun_out/ab_bench_err.txt | tail -1 | python -c '
import json,sys
try:
  j=json.loads(sys.stdin.readline())
  r=j["roofline"]
  print("qps", j["value"], "ms/step", j["ms_per_step"], "kernel_ms", r["kernel_ms_avg"], "host_ms", r["host_plan_ms"], "algo_frac", r["algorithmic_frac"], "exh_ms", j["roofline_other_mode"]["kernel_ms_avg"], "scored", r.get("docs_scored_per_launch"), "tasks", r.get("launch_tasks"), "kernels", r["kernel"], "parity", j["pruned_equals_exhaustive"], j["parity_checked_queries"])
except Exception as e:
  print("FAILED", e); print(open("gpurun_out/ab_bench_err.txt").read()[-1500:])'
  done
done
