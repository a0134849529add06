#!/bin/bash
# repeated headline runs with the host time of every enqueue call: where a slow run (1.5 instead of 1.15 ms per step)
# loses its time
for i in $(seq 1 ${RUNS:-8}); do
  BENCH_TRACE=1 python bench.py --workload ${W:-and2} --no-side --no-cpu-baseline --latency-queries 0 --no-pmc-inline --no-stream --steps 20 --warmup 5 > /tmp/hic.json 2> /tmp/hic.err
  python - <<'PY'
import json
j=json.loads(open('/tmp/hic.json').read().strip().splitlines()[-1])
marks=[l for l in open('/tmp/hic.err') if l.startswith('[bench] enqueue')]
print("ms/step %.3f host %.3f gpu %.3f | %s" % (j["ms_per_step"], j["roofline"]["host_plan_ms"], j["roofline"]["gpu_batch_ms"], marks[-1].strip()[8:] if marks else ""))
PY
done
