set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-queries 0 $EXTRA"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- $B > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/gpurun_out/prof_p1 -o p1 -- $B > $R/gpurun_out/prof_p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_p2 -o p2 -- $B > $R/gpurun_out/prof_p2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr -d $R/gpurun_out/prof_p3 -o p3 -- $B > $R/gpurun_out/prof_p3.log 2>&1
ls -R $R/gpurun_out | head -50
