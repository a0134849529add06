#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_xunion.py -x -q) > gpurun_out/c21_xu.log 2>&1
tail -2 gpurun_out/c21_xu.log
echo -n "or5 "; timeout 300 bash tools/quick.sh or5 --exhaustive 2>&1 | tail -1
echo -n "mixed "; timeout 300 bash tools/quick.sh mixed --exhaustive 2>&1 | tail -1
