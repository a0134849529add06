#!/usr/bin/env python
"""Per-query cost of pruned unions: 100 copies of each of the first N Zipf-sampled 5-term queries."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
dev.set_option("exhaustive", 0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ids = O.zipf_queries(40, 5, 256, seed=20260922)
rows = []
for q in ids:
    qs = [(O.MODE_OR, q.tolist())] * 100
    dev.prepare(qs)
    best = None
    for _ in range(2):
        dev.search_prepared(k)
        st = dev.last_batch_stats()
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    rows.append((best["kernel_ms"] * 10, sorted(q.tolist()), best["matches"] / 100, best["tiles"] // 100))
rows.sort(reverse=True)
for us, q, m, t in rows:
    print("%8.1f us/query  terms %-28s scored %9.0f tiles %d" % (us, q, m, t))
print("mean %.1f us" % np.mean([r[0] for r in rows]))
dev.close()
