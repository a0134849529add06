#!/bin/bash
# region timers of the shared-intersection kernel (a -DTQ_AS_TIMERS=1 variant build): wave cycles / 64 per region
#   python tools/build_variant.py ast tq_ashare.hip -DTQ_AS_TIMERS=1     (HERE, before gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_${VARIANT:-ast}.so
for ph in ${PHASES_LIST:-1 2 3 4 5 6 7 8 9}; do
  echo -n "region $ph: "
  TQ_DEBUG=$((ph<<16)) python tools/probe_ashare.py 2>&1 | tail -1 | grep -o 'kernel [0-9.]* ms.*counter [0-9.e+]*'
done
