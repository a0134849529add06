#!/bin/bash
# round 4: the profiles of the launches that include ashare_kernel, at the final kernel sources, then the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { tag=$1; w=$2; shift; shift; bash tools/profile_workload.sh $w $tag "$@" > gpurun_out/prof_$tag.log 2>&1; tail -1 gpurun_out/prof_$tag.log | cut -c1-120; }
run r04_and2 and2
KEY_SUFFIX=_t4096 run r04_and2_t4096 and2 --terms 4096
run r04_mixed mixed
KEY_SUFFIX=_s8 run r04_and2_s8 and2 --segments 8
