#!/usr/bin/env python
"""Count collector: bitmap words (tq_count.hip) against the exhaustive scan on the bench's query streams
(10M docs, 256 lists): wall time per batch of Searcher::search(&query, &Count)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
streams = {"or5": [(O.MODE_OR, q.tolist()) for q in O.zipf_queries(1000, 5, 256, seed=20260922)],
           "and2": [(O.MODE_AND, q.tolist()) for q in O.zipf_queries(10000, 2, 256, seed=20260921)]}
for name, qs in streams.items():
    for ratio in [int(x) for x in os.environ.get("PROBE_RATIOS", "32,0").split(",")]:
        dev.set_option("count_bitmap_ratio", ratio)
        ref = dev.count(qs)
        t0 = time.perf_counter()
        for _ in range(5):
            got = dev.count(qs)
        dt = (time.perf_counter() - t0) / 5
        st = dev.last_batch_stats()
        print("%-5s count_bitmap_ratio %2d: %8.3f ms per batch of %d (%s), sum %d" %
              (name, ratio, dt * 1e3, len(qs), "+".join(st["kernels"]), int(got.sum())))
dev.close()
