#!/bin/bash
# instruction counters of the shared-intersection kernel for a list of environments:
#   bash tools/pmc_ashare.sh "TQ_DEBUG=0" "TQ_DEBUG=1024" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf /tmp/pmc_as; 
  env $e timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_as -o p -- python $R/tools/probe_ashare.py > /tmp/pmc_as.log 2>&1
  tail -1 /tmp/pmc_as.log | cut -c1-150
  python - <<'PY'
import csv,glob,collections,re
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_as/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'ashare' in k or 'and_kernel' in k:
            per[re.search(r'(ashare_kernel|and_kernel)<[^>]*>', k).group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in per.items():
    print('  ',k,' '.join('%s=%.4g'%(c.replace('SQ_',''),sum(x)/len(x)) for c,x in sorted(v.items())), 'n=%d'%len(next(iter(v.values()))))
PY
done
