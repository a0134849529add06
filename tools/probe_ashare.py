#!/usr/bin/env python
"""Shared-intersection launch on the headline batch: kernel / step time and work counters
(TQ_DEBUG=32: (block, lead) pairs, 64: stage-C candidates, 256: blocks decoded, 0: docs that reached
the collector).  Knobs are read from the environment by the library (one process per setting)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

terms = int(os.environ.get("PROBE_TERMS", "256"))
nq = int(os.environ.get("PROBE_QUERIES", "10000"))
seg = O.synth_segment(10_000_000, n_terms=terms)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
ids = O.zipf_queries(nq, 2, terms, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
dev.prepare(qs)
for _ in range(3):
    dev.search_prepared(10)
st = dev.last_batch_stats()
t0 = time.perf_counter()
for _ in range(10):
    dev.search_prepared(10)
wall = (time.perf_counter() - t0) / 10
st = dev.last_batch_stats()
print("%s kernel %.3f ms host %.3f ms wall/step %.3f ms counter %.4g tasks %d kernels %s" %
      (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith(("TQ_", "PROBE_"))),
       st["kernel_ms"], st["host_plan_ms"], wall * 1e3, st["matches"], st["chunks"], st["kernels"]))
dev.close()
