#!/bin/bash
# A/B of the boolean shared launch against the union kernel + work counters
for v in ${VOCABS:-24}; do
  export PROBE_VOCAB=$v
  python tools/probe_bshare.py
  [ -n "$NOBASE" ] || TQ_BSHARE=0 python tools/probe_bshare.py
  for d in ${DEBUGS:-64}; do TQ_DEBUG=$d python tools/probe_bshare.py; done
  for sh in 0 1 2 3; do PROBE_SHAPE=$sh python tools/probe_bshare.py; done
done
