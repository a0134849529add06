#!/bin/bash
# parity of the shared-intersection launch + headline A/B in one box: bash tools/r4_ab.sh ["ENV=.. ENV=.." ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_ashare.py -x -q 2>&1 | tail -4; fi
for e in "$@"; do
  env $e timeout 300 python tools/probe_ashare.py 2>&1 | tail -1
done
