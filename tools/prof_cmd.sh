# usage: CMD="python bench.py ..." bash tools/prof_cmd.sh  -> kernel durations + basic PMC per kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_cmd
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
python $R/tools/read_prof.py $OUT/p1/p1_results.db | grep -v rocclr
