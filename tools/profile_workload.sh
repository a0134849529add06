#!/bin/bash
# rocprofv3 evidence for one bench workload (run on the GPU box through gpurun):
#   usage: bash tools/profile_workload.sh <workload> <tag> [extra bench args]
#   1. --kernel-trace --stats        -> per-kernel durations
#   2. --pmc passes (separate runs, never combined with traces) -> instruction mix, waits,
#      FETCH_SIZE / WRITE_SIZE, L2 hit rate
# Every pass runs under `timeout`: a rocprofv3 that dies on a counter set hangs until killed.
# Outputs land in gpurun_out/prof_<tag>/ (scratch); tools/summarize_profile.py turns them into the
# committed summaries — on the box into gpurun_out/summaries/, which tools/merge_summaries.py folds
# into profiles/ back home (in the order given: a later tag's traffic entries win).
W=${1:-and2}; TAG=${2:-r02_$W}; shift; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --latency-queries 0 --no-side --no-pmc-inline --no-stream $@"
echo "$B" > $OUT/command.txt
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -o p1 -- $B > $OUT/p1.log 2>&1
timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- $B > $OUT/p2.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p3 -- $B > $OUT/p3.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -o p4 -- $B > $OUT/p4.log 2>&1
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p5 -o p5 -- $B > $OUT/p5.log 2>&1
tail -1 $OUT/kt.log | cut -c1-300
SUMMARY_DIR=$R/gpurun_out/summaries python $R/tools/summarize_profile.py $OUT $TAG $W 10000000 "$KEY_SUFFIX"
# the raw per-dispatch CSVs stay on the box unless asked for (KEEP_RAW=1): gpurun merges at most 64 MiB back
if [ -z "$KEEP_RAW" ]; then rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/kt/kt_kernel_trace.csv $OUT/kt/*agent_info.csv; fi
