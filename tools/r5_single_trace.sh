#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_single
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/kt_single -o kt -- python $R/tools/r5_single.py 2>&1 | grep "batch"
python - <<'PY'
import csv,glob
for name in ('kernel_stats','memory_copy_stats'):
    for f in glob.glob('/tmp/kt_single/**/kt_%s.csv' % name, recursive=True):
        rows=list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print('%-80s calls %6s avg_us %8.1f total_ms %8.2f' % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
