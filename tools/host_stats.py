#!/usr/bin/env python
"""Distribution of the host planning time per step of a bench run: TQ_TRACE=1 lines on stdin."""
import re
import sys

rows = []
for line in sys.stdin:
    m = re.search(r"plan (\d+) us \(pre-pass (\d+) us, queries (\d+) us, chunks (\d+) us\), stage fill (\d+) us, enqueue (\d+)", line)
    if m:
        rows.append([int(x) for x in m.groups()])
    elif line.startswith("{"):
        import json
        j = json.loads(line)
        print("qps %.0f ms/step %.3f kernel %.3f host %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_avg"], j["roofline"]["host_plan_ms"]))
rows = rows[-40:]
names = ["plan", "pre-pass", "queries", "chunks", "stage fill", "enqueue"]
for i, n in enumerate(names):
    v = sorted(r[i] for r in rows)
    print("%-10s n=%d min %5d p50 %5d p90 %5d max %5d us" % (n, len(v), v[0], v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
