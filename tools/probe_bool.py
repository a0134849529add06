#!/usr/bin/env python
"""Boolean workload by query shape: kernel time and docs scored, pruned vs exhaustive."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402  (index generator only)
import tantivy_amd as T  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = T.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
ids = O.zipf_queries(500, 4, 256, seed=20260924)
M, S, N = T.MUST, T.SHOULD, T.MUST_NOT
shapes = {"+a +(b|c)": (3, [M, M, M], [0, 1, 1]), "+(a|b) +(c|d)": (4, [M, M, M, M], [0, 0, 1, 1]),
          "+a b -c": (3, [M, S, N], None), "+(a|b) +c": (3, [M, M, M], [0, 0, 1]),
          "+a b": (2, [M, S], None), "+a -b": (2, [M, N], None)}
for name, (nt, occ, cof) in shapes.items():
    qs = [(T.MODE_BOOL, q.tolist()[:nt], occ, cof, 0) for q in ids]
    out = []
    for ex in (1, 0):
        dev.set_option("exhaustive", ex)
        dev.prepare(qs)
        for _ in range(2):
            dev.search_prepared(10)
            st = dev.last_batch_stats()
        out.append(st)
    e, p = out
    print("%-16s exh %7.3f ms (scored %.3g) | pruned %7.3f ms (scored %.3g)" %
          (name, e["kernel_ms"], e["matches"], p["kernel_ms"], p["matches"]))
dev.close()
