#!/usr/bin/env python
"""Phrase kernel probe: the same term triples as 3-term AND (intersection cost) and as phrases."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256, with_positions=True, phrase_terms=32)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
rng = np.random.default_rng(20260923)
starts = rng.integers(0, 30, size=1000)


def run(name, qs, k=10):
    dev.prepare(qs)
    best = None
    for _ in range(3):
        dev.search_prepared(k)
        st = dev.last_batch_stats()
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    print("%-28s n=%4d kernel %8.3f ms  matches %.3g  chunks %d tiles %d" %
          (name, len(qs), best["kernel_ms"], best["matches"], best["chunks"], best["tiles"]))


run("AND3 same triples", [(O.MODE_AND, [int(s), int(s) + 1, int(s) + 2]) for s in starts])
run("PHRASE3", [(O.MODE_PHRASE, [int(s), int(s) + 1, int(s) + 2]) for s in starts])
run("PHRASE3 (0,1,2) x200", [(O.MODE_PHRASE, [0, 1, 2])] * 200)
run("PHRASE3 (27,28,29) x200", [(O.MODE_PHRASE, [27, 28, 29])] * 200)
run("PHRASE2 (0,1) x200", [(O.MODE_PHRASE, [0, 1])] * 200)
dev.close()
