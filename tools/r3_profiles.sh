#!/bin/bash
# round 3 evidence: kernel traces + PMC passes of every bench workload (tools/profile_workload.sh);
# back home: python tools/merge_summaries.py r03_and2_exhaustive r03_or5_exhaustive r03_mixed_exhaustive r03_and2 r03_or5 r03_phrase3 r03_mixed r03_bool r03_mixed_s8
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for w in and2 or5 phrase3 mixed bool; do
  bash tools/profile_workload.sh $w r03_$w > gpurun_out/prof_$w.log 2>&1; tail -1 gpurun_out/prof_$w.log | cut -c1-100
done
bash tools/profile_workload.sh and2 r03_and2_exhaustive --exhaustive > gpurun_out/prof_and2x.log 2>&1
bash tools/profile_workload.sh or5 r03_or5_exhaustive --exhaustive > gpurun_out/prof_or5x.log 2>&1
bash tools/profile_workload.sh mixed r03_mixed_exhaustive --exhaustive > gpurun_out/prof_mixedx.log 2>&1
KEY_SUFFIX=_s8 bash tools/profile_workload.sh mixed r03_mixed_s8 --segments 8 > gpurun_out/prof_mixed_s8.log 2>&1; tail -1 gpurun_out/prof_mixed_s8.log | cut -c1-100
