#!/usr/bin/env python
"""Small synchronous batches of the bench's boolean shapes: per-batch wall time by batch size under the
environment's knobs (TQ_BSHARE=0: the per-query union kernel)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd as T  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = T.DeviceIndex([seg], devices=[0])
ids = O.zipf_queries(2048, 4, 256, seed=20260924)
M, S, N = T.MUST, T.SHOULD, T.MUST_NOT
shapes = [(3, [M, M, M], [0, 1, 1]), (4, [M, M, M, M], [0, 0, 1, 1]), (3, [M, S, N], None), (3, [M, M, M], [0, 0, 1])]
qs = []
for i, q in enumerate(ids):
    nt, occ, cof = shapes[i % 4]
    qs.append((T.MODE_BOOL, q.tolist()[:nt], occ, cof, 0))
dev.set_option("exhaustive", 0)
dev.prepare(qs)
dev.search_prepared(10)  # (probe tables of every list built once)
row = []
for b in (1, 4, 16, 64, 256, 1024, 2048):
    dev.prepare(qs[:b])
    t = []
    for _ in range(43):
        t1 = time.perf_counter()
        dev.search_prepared(10)
        t.append(time.perf_counter() - t1)
    t = sorted(t[3:])
    row.append("%d: p50 %.3f ms %s" % (b, t[len(t) // 2] * 1e3, "+".join(dev.last_batch_stats()["kernels"])))
print(" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("TQ_")), "|", " | ".join(row))
dev.close()
