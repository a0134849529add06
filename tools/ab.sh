#!/bin/bash
# A/B of library variants on the GPU box: bash tools/ab.sh "<variant names|base>" "<TQ_DEBUG values>" <k> 
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in $1; do
  for d in $2; do
    if [ "$v" = base ]; then unset TQ_LIB_PATH; else export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_$v.so; fi
    echo -n "$v "; TQ_DEBUG=$d python $R/tools/probe_or3.py ${3:-100} 2>&1 | tail -1
  done
done
