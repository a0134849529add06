#!/bin/bash
# rocprofv3 kernel trace of the encoder bench -> gpurun_out/prof_enc (summary copied to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_enc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/tools/bench_encode.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log | cut -c1-200
head -12 $OUT/kt/kt_kernel_stats.csv
