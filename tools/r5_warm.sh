#!/bin/bash
# the warm-up launch of the shared intersections: how much of every leader goes out first, in how long tasks
for cfg in "X=0" "TQ_AS_WARM_PERMILLE=0" "TQ_AS_WARM_PERMILLE=1" "TQ_AS_WARM_PERMILLE=4" "TQ_AS_WARM_BLOCKS=1" "TQ_AS_WARM_BLOCKS=4 TQ_AS_WARM_PERMILLE=4" "X=1"; do
  echo "$cfg: $(env $cfg bash tools/quick2.sh and2 --no-pmc-inline --no-stream)"
done
