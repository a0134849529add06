#!/bin/bash
# kernel + memory-copy trace of the pipelined bench loop: timeline of the last two steps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_b
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt_b -o kt -- python $R/bench.py --workload ${W:-and2} --no-side --no-cpu-baseline --latency-queries 0 --steps 6 --warmup 2 "$@" > /tmp/kt_b.log 2>&1
tail -1 /tmp/kt_b.log | cut -c1-200
python - <<'PY'
import csv,glob,re
ev=[]
for f in glob.glob('/tmp/kt_b/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        m=re.search(r'(\w+_kernel\w*|__amd_rocclr_\w+)', n)
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else n[:40]))
for f in glob.glob('/tmp/kt_b/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','')+' '+r.get('Bytes', r.get('Size',''))))
ev.sort()
# last 60 events
t0=ev[-60][0]
for a,b,k in ev[-60:]:
    print('%10.1f %10.1f  %8.1f us  %s' % ((a-t0)/1e3,(b-t0)/1e3,(b-a)/1e3,k))
PY
