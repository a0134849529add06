#!/bin/bash
# doc-major union launch: parity tests, then the exhaustive or5 / mixed timings, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_xunion.py -x -q) > gpurun_out/c16_xu.log 2>&1
tail -15 gpurun_out/c16_xu.log
timeout 300 bash tools/quick.sh or5 2>&1 | tail -1
timeout 300 bash tools/quick.sh mixed 2>&1 | tail -1
echo "--- bool base / u4"
timeout 300 bash tools/quick.sh bool 2>&1 | tail -1
TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_u4.so timeout 300 bash tools/quick.sh bool 2>&1 | tail -1
(timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/c16_tests.log 2>&1
tail -3 gpurun_out/c16_tests.log
