#!/bin/bash
# memory-side counters of the shared-intersection kernel: bash tools/pmc_ashare2.sh "ENV=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmc_as
  env $e timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_as -o p -- python $R/tools/probe_ashare.py > /tmp/pmc_as.log 2>&1
  python - <<'PY'
import csv,glob,collections,re
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_as/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        m=re.search(r'(ashare_kernel|and_kernel)<[^>]*>', k)
        if m: per[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in per.items():
    # ashare: two launches per batch (warm-up, main): print both halves
    for c,x in sorted(v.items()):
        if 'ashare' in k:
            print('  %-28s %-16s warm-up %.4g  main %.4g' % (k, c, sum(x[0::2])/len(x[0::2]), sum(x[1::2])/len(x[1::2])))
        else:
            print('  %-28s %-16s %.4g' % (k, c, sum(x)/len(x)))
PY
  done
done
