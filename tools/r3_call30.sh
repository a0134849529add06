#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/c30_tests.log 2>&1
grep -E "passed|failed|Error" gpurun_out/c30_tests.log | tail -3
for w in and2 mixed; do TQ_TRACE=1 python bench.py --workload $w --no-side --no-cpu-baseline --latency-queries 0 --steps 30 --warmup 3 2> gpurun_out/tr.err | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j[\"roofline\"]
print(\"qps\", j[\"value\"], \"ms/step\", j[\"ms_per_step\"], \"kernel_ms\", r[\"kernel_ms_avg\"], \"host_plan_ms\", r[\"host_plan_ms\"], j[\"pruned_equals_exhaustive\"])"; grep "\[tq\]" gpurun_out/tr.err | tail -2; done
