#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_xunion.py -x -q) > gpurun_out/c18_xu.log 2>&1
tail -3 gpurun_out/c18_xu.log
for v in base xb8; do
  if [ "$v" = base ]; then unset TQ_LIB_PATH; else export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_$v.so; fi
  echo -n "$v "; timeout 300 bash tools/quick.sh or5 --exhaustive 2>&1 | tail -1
done
unset TQ_LIB_PATH
echo -n "mixed "; timeout 300 bash tools/quick.sh mixed --exhaustive 2>&1 | tail -1
export TQ_LIB_PATH=$R/tantivy_amd/lib/variants/libtantivy_amd_xt.so
for ph in 1 2 3 4; do
  echo -n "timer $ph: "; TQ_DEBUG=$((ph<<16)) timeout 300 bash tools/quick.sh or5 --exhaustive 2>&1 | tail -1 | grep -o "kernel_ms [0-9.]*\|scored [0-9]*" | tr '\n' ' '; echo
done
