#!/bin/bash
# boolean shared launch: knob sweep on the bench workload
bash tools/quick2.sh bool
TQ_TRACE=1 python bench.py --workload bool --no-side --no-cpu-baseline --latency-queries 0 --steps 2 --warmup 1 2>&1 | grep "tq. group" | tail -2
for e in "TQ_AS_TASK_PAIRS=256" "TQ_AS_TASK_PAIRS=1024" "TQ_BOOL_OPT_LEAD=0" "TQ_AS_WARM_PERMILLE=5" "TQ_AS_WARM_PERMILLE=10 TQ_AS_WARM_BLOCKS=4" "TQ_AS_GROUP=16"; do
  echo "$e"; env $e bash tools/quick2.sh bool
done
