#!/bin/bash
# extra PMC passes for one workload (vector-memory write instructions = scratch spills, L2 request mix)
W=${1:-or5}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_extra_$W
rm -rf $OUT; mkdir -p $OUT
B="python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --latency-queries 0 --no-side"
timeout 120 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/e1 -o e1 -- $B > $OUT/e1.log 2>&1
timeout 120 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum --output-format csv -d $OUT/e2 -o e2 -- $B > $OUT/e2.log 2>&1
python - <<PY
import csv, collections
for e in ("e1","e2"):
    per=collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open("$OUT/%s/%s_counter_collection.csv"%(e,e))):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as ex:
        print(e, "failed", ex); print(open("$OUT/%s.log"%e).read()[-600:]); continue
    for k in per:
        if "union_kernel" in k or "and_kernel<1, true" in k or "phrase_kernel" in k:
            print(k[:70], {c: round(sum(v)/len(v)) for c,v in per[k].items()})
PY
