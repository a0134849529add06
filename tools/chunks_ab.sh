for c in ${CHUNK_LIST:-131072 65536 32768}; do echo -n "chunks $c: "; TQ_CHUNKS=$c python bench.py --workload and2 --no-side --no-cpu-baseline --latency-queries 0 --steps ${STEPS:-20} --warmup 3 2>&1 | tail -1 | python -c '
import json,sys
j=json.loads(sys.stdin.readline()); r=j["roofline"]
print("qps", j["value"], "ms/step", j["ms_per_step"], "kernel_ms", r["kernel_ms_avg"], "host_plan_ms", r["host_plan_ms"])'; done
