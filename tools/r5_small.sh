#!/bin/bash
# small synchronous batches: tile / chunk size of the per-query kernels for the smallest launches
# (TQ_AND_SMALL_BLOCKS=0 = 64-block tiles and 128-unit chunks whatever the batch, the behaviour before), two-level merge
for cfg in "TQ_AND_SMALL_BLOCKS=0" "TQ_AND_SMALL_BLOCKS=8192" "TQ_AND_SMALL_BLOCKS=16384" "TQ_AND_SMALL_BLOCKS=8192 TQ_AND_MIN_TILE=2" "TQ_AND_SMALL_BLOCKS=8192 TQ_AND_MIN_TILE=8"; do
  env $cfg BATCHES=1,2,4,8,16,64 python tools/latency_ab.py 2>&1 | tail -1
done
