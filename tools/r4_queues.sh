#!/bin/bash
# per-XCD task queues of the shared launches against one queue
TQ_DEBUG=32768 python tools/probe_ashare.py 2>&1 | tail -1
for q in 8 1 8; do
  export TQ_AS_QUEUES=$q
  echo "TQ_AS_QUEUES=$q"
  bash tools/quick2.sh and2
done
