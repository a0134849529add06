#!/bin/bash
# the default bench line (what the driver runs) + its wall time
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{ time python bench.py > gpurun_out/r6_bench_line.json 2> gpurun_out/r6_bench_err.txt ; } 2> gpurun_out/r6_bench_time.txt
tail -c 600 gpurun_out/r6_bench_err.txt; cat gpurun_out/r6_bench_time.txt
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r6_bench_line.json").read().strip().splitlines()[-1])
r=j["roofline"]
print("HEADLINE", j["value"], "q/s  distinct", j.get("value_distinct"), " ms/step", j["ms_per_step"], "kernel", r["kernel_ms_avg"], "host", r["host_plan_ms"], "frac", r["frac"], "traffic", r["traffic"], "p50", j["p50_latency_ms"])
print("cpu", j["cpu_baseline"]["value"] if j["cpu_baseline"] else None)
for k,v in (j["other_workloads"] or {}).items():
    print(" ", k, "qps", v["qps"], "ms/step", v["ms_per_step"], "kernel", v["kernel_ms_avg"], "exh", v["exhaustive_kernel_ms"], "frac", v["roofline_frac"], "algo", v["algorithmic_frac"], "host", v["host_plan_ms"], v["kernels"], "checked", v["parity_checked_queries"])
s=j["strong_scaling"]
if s: print(" strong", s["qps"], "ms/step", s["ms_per_step"], "kernel", s["kernel_ms_per_gpu"], "host", s["host_plan_ms_per_gpu"])
for k,v in ((j.get("stream") or {}).get("by_terms") or {}).items():
    print(" stream", k, "cold", v["cold_first_batch_ms"], "steady", v["steady_qps"], "second", v["second_pass_qps"], "derived_x", v["derived_x"])
lc=j.get("latency_curve") or {}
print(" threads", {k:(v["qps"],v["p50_ms"],v["p99_ms"]) for k,v in (lc.get("threads") or {}).items()})
PY
