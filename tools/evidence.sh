#!/bin/bash
# round 6 evidence: kernel traces + PMC passes (tools/profile_workload.sh) of every bench workload at the final
# kernels, the GPU test log, smoke(), the default bench line.  PART=1 profiles, PART=2 tests + smoke + bench.
#   back home: python tools/merge_summaries.py r06_and2 r06_and2_distinct r06_and2_s8 r06_and2_t4096 r06_bool r06_mixed r06_or5 r06_phrase3 r06_phrase3_adj
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
run() { tag=$1; w=$2; shift; shift; bash tools/profile_workload.sh $w $tag "$@" > gpurun_out/prof_$tag.log 2>&1; tail -1 gpurun_out/prof_$tag.log | cut -c1-120; }
if [ "${PART:-1}" = 1 ]; then
  run r06_and2 and2
  run r06_and2_distinct and2_distinct
  KEY_SUFFIX=_s8 run r06_and2_s8 and2 --segments 8
  KEY_SUFFIX=_t4096 run r06_and2_t4096 and2 --terms 4096
  run r06_bool bool
  run r06_mixed mixed
  run r06_or5 or5
  run r06_phrase3 phrase3
  run r06_phrase3_adj phrase3_adj
else
  ( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > gpurun_out/r06_gpu_tests.log 2>&1
  cat gpurun_out/r06_gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r06_smoke.log
  python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
  tail -c 600 gpurun_out/r06_bench_line.json
fi
