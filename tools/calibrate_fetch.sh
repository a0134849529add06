#!/bin/bash
# FETCH_SIZE calibration on this path's access patterns (run on the GPU box through gpurun).
# Builds tools/calib/fetch_calib.bin if missing, runs it under rocprofv3 --pmc (one pass per
# counter group, no traces) and writes profiles/fetch_calibration.json via calibrate_fetch.py.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_calib
rm -rf $OUT; mkdir -p $OUT
BIN=$R/tools/calib/fetch_calib.bin
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 $R/tools/calib/fetch_calib.hip -o $BIN
$BIN > $OUT/plain.jsonl 2> $OUT/plain.err
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA_RDREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u > $OUT/rdreq_counters.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c1 -o c1 -- $BIN > $OUT/c1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/c2 -o c2 -- $BIN > $OUT/c2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c3 -o c3 -- $BIN > $OUT/c3.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $OUT/c4 -o c4 -- $BIN > $OUT/c4.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/c5 -o c5 -- $BIN > $OUT/c5.log 2>&1
python $R/tools/calibrate_fetch.py $OUT
