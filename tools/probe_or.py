#!/usr/bin/env python
"""OR kernel probe: fixed per-window cost vs posting work (HIP-event kernel time)."""
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


def run(name, qs, k=100):
    for ow in (0, 1):
        dev.set_option("or_windows", ow)
        _run(("cand " if ow == 0 else "wind ") + name, qs, k)


def _run(name, qs, k=100):
    out = []
    for ex in (1, 0):
        dev.set_option("exhaustive", ex)
        dev.prepare(qs)
        best = None
        for _ in range(2):
            dev.search_prepared(k)
            st = dev.last_batch_stats()
            best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
        out.append(best)
    e, p = out
    print("%-28s n=%4d exh %8.3f ms (scored %.3g) | pruned %8.3f ms (scored %.3g) chunks %d tiles %d" %
          (name, len(qs), e["kernel_ms"], e["matches"], p["kernel_ms"], p["matches"], e["chunks"],
           e["tiles"]))


run("sparse5 (251..255)", [(O.MODE_OR, [251, 252, 253, 254, 255])] * 200)
run("dense5 (0..4)", [(O.MODE_OR, [0, 1, 2, 3, 4])] * 200)
run("mixed (19,7,10,5,237)", [(O.MODE_OR, [19, 7, 10, 5, 237])] * 200)
run("mixed k=10", [(O.MODE_OR, [19, 7, 10, 5, 237])] * 200, k=10)
ids = O.zipf_queries(200, 5, 256, seed=20260922)
run("zipf 200", [(O.MODE_OR, q.tolist()) for q in ids])
dev.close()
