#!/bin/bash
# which phase of tq_search_batch_device holds a 7 ms stall (TQ_TRACE lines of slow calls)
for i in $(seq 1 ${RUNS:-8}); do
  TQ_TRACE=1 BENCH_TRACE=1 python bench.py --workload ${W:-and2_distinct} --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --no-stream --steps 20 --warmup 5 > /tmp/hic.json 2> /tmp/hic.err
  python - <<'PY'
import json,re
j=json.loads(open('/tmp/hic.json').read().strip().splitlines()[-1])
print("ms/step %.3f" % j["ms_per_step"])
lines=[l.strip() for l in open('/tmp/hic.err') if l.startswith('[tq] plan')]
for idx,l in enumerate(lines):
    nums=[int(x) for x in re.findall(r'(\d+) us', l)]
    if max(nums[:1]+nums[4:]) > 3000 and idx > 8:
        print("  call %d of %d: %s" % (idx, len(lines), l[:200]))
PY
done
