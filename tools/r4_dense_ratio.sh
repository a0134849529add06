#!/bin/bash
# every list of the 256-term vocabulary with a bitmap (dense_ratio 512) against the default (128)
for r in 128 512; do
  export TQ_OPT_dense_ratio=$r
  echo "dense_ratio $r"
  for w in ${WORKLOADS:-bool and2 or5 mixed phrase3}; do bash tools/quick2.sh $w; done
done
