#!/bin/bash
# round 4, GPU call 1: parity of the shared-intersection launch, then the headline A/B in one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ashare.py -x -q 2>&1 | tail -25 > gpurun_out/r4c1_tests.log
cat gpurun_out/r4c1_tests.log
for a in 1 0; do
  echo "TQ_ASHARE=$a"
  TQ_ASHARE=$a TQ_TRACE=0 timeout 600 bash tools/quick.sh and2 2>&1 | tail -3
done | tee gpurun_out/r4c1_ab.log
