#!/bin/bash
# round 5 evidence: kernel traces + PMC passes (tools/profile_workload.sh) of every bench workload at the final
# kernels, the GPU test log, smoke(), the default bench line.  PART=1 profiles, PART=2 tests + bench.
#   back home: python tools/merge_summaries.py r05_and2 r05_and2_distinct r05_and2_t4096 r05_bool r05_mixed r05_or5 r05_phrase3
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { tag=$1; w=$2; shift; shift; bash tools/profile_workload.sh $w $tag "$@" > gpurun_out/prof_$tag.log 2>&1; tail -1 gpurun_out/prof_$tag.log | cut -c1-120; }
if [ "${PART:-1}" = 1 ]; then
  run r05_and2 and2
  run r05_and2_distinct and2_distinct
  KEY_SUFFIX=_t4096 run r05_and2_t4096 and2 --terms 4096
  run r05_bool bool
  run r05_mixed mixed
  run r05_or5 or5
  run r05_phrase3 phrase3
else
  ( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r05_gpu_tests.log 2>&1
  cat gpurun_out/r05_gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r05_smoke.log
  python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench.err
  tail -c 600 gpurun_out/r05_bench_line.json
fi
