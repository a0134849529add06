#!/bin/bash
# round 4 evidence: kernel traces + PMC passes (tools/profile_workload.sh), in two gpurun calls (PART=1 / 2);
# back home: python tools/merge_summaries.py r04_and2 r04_and2_t4096 r04_bool r04_mixed r04_and2_s8 r04_or5 r04_phrase3
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { tag=$1; w=$2; shift; shift; bash tools/profile_workload.sh $w $tag "$@" > gpurun_out/prof_$tag.log 2>&1; tail -1 gpurun_out/prof_$tag.log | cut -c1-120; }
if [ "${PART:-1}" = 1 ]; then
  run r04_and2 and2
  KEY_SUFFIX=_t4096 run r04_and2_t4096 and2 --terms 4096
  run r04_bool bool
else
  run r04_mixed mixed
  KEY_SUFFIX=_s8 run r04_and2_s8 and2 --segments 8
  run r04_or5 or5
  run r04_phrase3 phrase3
fi
if [ "${PART:-1}" = 2 ]; then  # work counters of the headline batch
  for d in 0 32 64 256 512 4096 8192 16384; do TQ_DEBUG=$d python tools/probe_ashare.py 2>&1 | tail -1; done > gpurun_out/r04_ashare_counters.txt
  cat gpurun_out/r04_ashare_counters.txt
fi
