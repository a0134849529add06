#!/bin/bash
# the stream block alone (fresh batches, Query::weight timed): serial against a preparing thread one batch ahead
for flag in "--stream-serial" ""; do
  python bench.py --workload and2 --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --steps 5 --warmup 2 --stream-vocabs ${VOCABS:-256,4096} $flag 2>/dev/null | tail -1 | python -c '
import json,sys
j=json.loads(sys.stdin.readline())
for k,v in j["stream"]["by_terms"].items():
    print("%-16s terms %6s: steady %9.0f q/s, %.3f ms/batch, prepare %.3f ms (%s), host_plan %.3f, kernel %.3f, parity %s; second pass %9.0f q/s %.3f ms/batch; loop %s derived_x %.2f" % (sys.argv[1] or "overlapped", k, v["steady_qps"], v["steady_ms_per_batch"], v["prepare_ms_per_batch"], v.get("prepare_thread"), v["host_plan_ms"], v["kernel_ms_avg"], v["parity_checked_queries"], v["second_pass_qps"], v["second_pass_ms_per_batch"], list(v["loop_ms_per_batch"].values()), v["derived_bytes"] / v["tantivy_bytes"]))' "$flag"
done
