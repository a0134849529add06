#!/bin/bash
# region timers of the shared-union kernel (a -DTQ_US_TIMERS=1 variant build): wave cycles / 64 per region
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_ust.so
W=${1:-or5}; shift
for ph in ${PHASES_LIST:-1 2 3 4 5 6 7 8 9 10 11 12 13}; do
  echo -n "region $ph: "
  TQ_DEBUG=$((ph<<16)) python bench.py --workload $W --no-side --no-cpu-baseline --latency-queries 0 --steps 2 --warmup 1 "$@" 2>/dev/null | tail -1 | python -c '
import json,sys
j=json.loads(sys.stdin.readline()); print(j["roofline"]["docs_scored_per_launch"], "kernel_ms", j["roofline"]["kernel_ms_avg"])' 
done
