#!/bin/bash
export PROBE_VOCAB=24
for ol in 1 0; do
  export TQ_BOOL_OPT_LEAD=$ol
  PROBE_SHAPE=2 python tools/probe_bshare.py
  PROBE_SHAPE=2 TQ_DEBUG=64 python tools/probe_bshare.py
  PROBE_SHAPE=2 TQ_DEBUG=256 python tools/probe_bshare.py
  PROBE_SHAPE=2 TQ_DEBUG=32 python tools/probe_bshare.py
done
