#!/bin/bash
# rocprofv3 evidence for the default bench command (run on the GPU box through gpurun):
#   1. --kernel-trace --stats        -> per-kernel durations
#   2. --pmc passes (separate runs)  -> instruction mix, waits, HBM FETCH_SIZE / WRITE_SIZE
# Outputs land in gpurun_out/prof_r01/ (scratch); tools/summarize_profile.py turns them into the
# committed summaries under profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-queries 0 $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -o p1 -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p3 -- $B > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -o p4 -- $B > $OUT/p4.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p5 -o p5 -- $B > $OUT/p5.log 2>&1
tail -1 $OUT/kt.log | cut -c1-300
find $OUT -name "*.csv" | head -30
