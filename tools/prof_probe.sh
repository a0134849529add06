# usage: CASE="0,1" bash tools/prof_probe.sh   -> PMC counters of the AND kernel for one probe case
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/tools/probe.py 10000000 $CASE"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM -d $R/gpurun_out/pp1 -o p1 -- $B > $R/gpurun_out/pp1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/pp2 -o p2 -- $B > $R/gpurun_out/pp2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC -d $R/gpurun_out/pp3 -o p3 -- $B > $R/gpurun_out/pp3.log 2>&1
grep "1000 x\|^all" $R/gpurun_out/pp1.log
python $R/tools/read_prof.py $R/gpurun_out/pp1/p1_results.db $R/gpurun_out/pp2/p2_results.db $R/gpurun_out/pp3/p3_results.db | grep "and_kernel"
