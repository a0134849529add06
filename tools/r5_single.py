#!/usr/bin/env python
"""One query per call, synchronous, 300 calls: what a launch costs at its floor (kernel / copy trace under rocprofv3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
ids = O.zipf_queries(400, 2, 256, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
dev.search(qs, 10)  # every term prepared
n = int(os.environ.get("BATCH", "1"))
t = []
for i in range(300):
    dev.prepare(qs[(i * n) % 300:(i * n) % 300 + n])
    t1 = time.perf_counter()
    dev.search_prepared(10)
    t.append(time.perf_counter() - t1)
t = sorted(t[20:])
print("batch %d: p50 %.3f ms p10 %.3f ms" % (n, t[len(t) // 2] * 1e3, t[len(t) // 10] * 1e3))
dev.close()
