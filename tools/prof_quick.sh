#!/bin/bash
# two PMC passes (instruction mix, busy / wait cycles) + a kernel trace of one workload: bash tools/prof_quick.sh <workload> <tag> [bench args]
W=$1; TAG=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --latency-queries 0 --no-side $@"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/p1 -o p1 -- $B > $OUT/p1.log 2>&1
timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- $B > $OUT/p2.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS --output-format csv -d $OUT/p3 -o p3 -- $B > $OUT/p3.log 2>&1
grep -h "xunion\|ushare" $OUT/kt/kt_kernel_stats.csv | cut -c1-160
