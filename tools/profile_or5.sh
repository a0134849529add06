cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_or5
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --workload or5 --queries 1000 --steps 5 --warmup 1 --no-cpu-baseline --latency-queries 0 > $OUT/kt.log 2>&1
head -8 $OUT/kt/kt_kernel_stats.csv | cut -c1-160
