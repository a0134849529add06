#!/bin/bash
for t in 256 4096 65536; do
  for p in 1 0; do
    echo "terms $t TQ_AS_PROBE=$p"; TQ_AS_PROBE=$p bash tools/quick2.sh and2 --terms $t
  done
done
