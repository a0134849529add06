#!/bin/bash
# union kernel diagnostics on the GPU box: work counters and phase timers (or5, k=100 and k=10)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for k in 100 10; do
  for d in 0 32 64 128 256; do TQ_DEBUG=$d python tools/probe_or3.py $k 2>&1 | tail -1; done
done
python tools/probe_phases.py or5 2>&1 | tail -8
