#!/usr/bin/env python
"""Union kernel work counters (TQ_DEBUG=32 counts leader blocks reaching stage A, 64 candidates
reaching stage B, 128 tiles that were not skipped as non-essential; 0 = docs scored)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
ids = O.zipf_queries(1000, 5, 256, seed=20260922)
qs = [(O.MODE_OR, q.tolist()) for q in ids]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev.set_option("exhaustive", 0)
dev.prepare(qs)
for _ in range(2):
    dev.search_prepared(k)
    st = dev.last_batch_stats()
print("TQ_DEBUG=%s k=%d kernel %.3f ms counter %.4g tiles %d chunks %d" %
      (os.environ.get("TQ_DEBUG", "0"), k, st["kernel_ms"], st["matches"], st["tiles"], st["chunks"]))
dev.close()
