#!/bin/bash
# kernel trace of the headline batch (probe_ashare.py): per-kernel durations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
rm -rf /tmp/kt_as
env $e timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_as -o kt -- python $R/tools/probe_ashare.py > /tmp/kt_as.log 2>&1
tail -1 /tmp/kt_as.log | cut -c1-170
python - <<'PY'
import csv,glob,re,collections
f=glob.glob('/tmp/kt_as/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
per=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    m=re.search(r'(ashare_kernel|and_kernel|merge_lists_kernel|merge_kernel|merge_segments_kernel)(<[^>]*>)?', n)
    if not m: continue
    per[m.group(0)].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k,v in per.items():
    d=[(b-a)/1e3 for a,b in v]
    d2=d[len(d)//2:]
    print('  %-40s n=%d  last-half avg %.1f us  min %.1f max %.1f' % (k, len(d), sum(d2)/len(d2), min(d2), max(d2)))
# timeline of the last batch: ashare launches and and_kernel
last=sorted([(a,b,k) for k,v in per.items() for a,b in v[-2:]])
t0=last[0][0]
for a,b,k in last[-8:]:
    print('    %-40s start %.1f us  end %.1f us' % (k,(a-t0)/1e3,(b-t0)/1e3))
PY
done
