#!/usr/bin/env python
"""T host threads x single-query Searcher::search calls (tq_search_one coalescing) on the headline
stream: throughput, p50 / p99 latency and queries per launch by arrival window (submit_window_us)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
ids = O.zipf_queries(10000, 2, 256, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
for window in [int(x) for x in os.environ.get("WINDOWS", "100,0,30,300").split(",")]:
    dev.set_option("submit_window_us", window)
    for nthreads in [int(x) for x in os.environ.get("THREADS", "16,64,256").split(",")]:
        n = min(len(qs), nthreads * 100)
        dev.search_concurrent(qs[:4 * nthreads], 10, nthreads)
        dev.submit_stats(reset=True)
        _, _, _, _, lat_ms, wall_ms = dev.search_concurrent(qs[:n], 10, nthreads)
        st = dev.submit_stats()
        v = sorted(float(x) for x in lat_ms)
        print("window %4d us threads %4d: %8.0f q/s p50 %.3f ms p99 %.3f ms, %.1f queries per launch (max %d)" %
              (window, nthreads, n / (wall_ms * 1e-3), v[len(v) // 2], v[int(len(v) * 0.99)],
               st["queries"] / max(1, st["batches"]), st["max_batch"]))
dev.close()
