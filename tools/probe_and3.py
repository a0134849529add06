#!/usr/bin/env python
"""AND kernel work counters on the bench mix (TQ_DEBUG=32: leader blocks reaching stage A,
64: candidates reaching stage B, 128: candidates reaching stage C; 0: docs scored)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
ids = O.zipf_queries(10000, 2, 256, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
dev.prepare(qs)
for _ in range(2):
    dev.search_prepared(10)
    st = dev.last_batch_stats()
print("TQ_DEBUG=%s kernel %.3f ms counter %.4g tiles %d chunks %d" %
      (os.environ.get("TQ_DEBUG", "0"), st["kernel_ms"], st["matches"], st["tiles"], st["chunks"]))
dev.close()
