#!/usr/bin/env python
"""Small synchronous batches of the headline stream (one caller, plan + H2D + kernels + D2H per call):
per-batch wall time by batch size — with the knobs of the environment (TQ_ASHARE=0, TQ_AS_MIN_LEADS=...)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
ids = O.zipf_queries(10000, 2, 256, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
row = []
for b in [int(x) for x in os.environ.get("BATCHES", "1,16,64,256,1024,4096").split(",")]:
    dev.prepare(qs[:b])
    t = []
    for _ in range(63):
        t1 = time.perf_counter()
        dev.search_prepared(10)
        t.append(time.perf_counter() - t1)
    t = sorted(t[3:])
    row.append("%d: p50 %.3f ms %s" % (b, t[len(t) // 2] * 1e3, "+".join(dev.last_batch_stats()["kernels"])))
print(" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("TQ_")), "|", " | ".join(row))
dev.close()
