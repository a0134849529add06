#!/bin/bash
# kernel trace of a short bench run: per-kernel durations (what a step spends outside the scan kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${W:-and2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$W
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$W -o kt -- python $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --latency-queries 0 --no-side > /tmp/kt_$W.log 2>&1
tail -1 /tmp/kt_$W.log | cut -c1-200
python - <<PY
import csv,glob
f=glob.glob('/tmp/kt_$W/**/kt_kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print('%-90s calls %6s total_us %10.1f avg_us %9.1f' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
PY
