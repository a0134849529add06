#!/usr/bin/env python
"""Where does the AND kernel's time go?  Builds the bench index once and times homogeneous query
batches (HIP-event kernel time from tq_last_batch_stats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402  (index generator only)
import tantivy_amd  # noqa: E402


def main():
    docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    seg = O.synth_segment(docs, n_terms=256)
    dev = tantivy_amd.DeviceIndex([seg], devices=[0])
    dev.set_option("timing", 1)
    ids = O.zipf_queries(10000, 2, 256, seed=20260921)

    def run(name, qs, k=10, reps=3):
        if not qs:
            return
        res = []
        for ex in (1, 0):
            dev.set_option("exhaustive", ex)
            dev.prepare(qs)
            best = None
            for _ in range(reps):
                dev.search_prepared(k)
                st = dev.last_batch_stats()
                best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
            res.append(best)
        e, p = res
        print("%-34s n=%5d  exh %8.3f ms %7.1f GB/s | pruned %8.3f ms %7.1f GB/s  chunks %d/%d tiles %d" %
              (name, len(qs), e["kernel_ms"], e["algorithmic_bytes"] / e["kernel_ms"] / 1e6,
               p["kernel_ms"], e["algorithmic_bytes"] / p["kernel_ms"] / 1e6, e["chunks"], p["chunks"],
               e["tiles"]))

    if len(sys.argv) > 2:  # single case, for rocprofv3: "a,b" or "all"
        if sys.argv[2] == "all":
            run("all", [(O.MODE_AND, q.tolist()) for q in ids], reps=2)
        else:
            a, b = [int(x) for x in sys.argv[2].split(",")]
            run("1000 x (%d,%d)" % (a, b), [(O.MODE_AND, [a, b])] * 1000, reps=2)
        dev.close()
        return
    allq = [(O.MODE_AND, q.tolist()) for q in ids]
    run("all (bench mix)", allq)
    mn = ids.min(axis=1)
    mx = ids.max(axis=1)
    for lo, hi in [(0, 1), (1, 4), (4, 16), (16, 32), (32, 64), (64, 256)]:
        sel = [(O.MODE_AND, q.tolist()) for q, a in zip(ids, mn) if lo <= a < hi]
        run("driver rank in [%d,%d)" % (lo, hi), sel)
    for lo, hi in [(0, 4), (4, 16), (16, 64), (64, 256)]:
        sel = [(O.MODE_AND, q.tolist()) for q, a, b in zip(ids, mn, mx) if a < 16 and lo <= b < hi]
        run("dense driver, leader in [%d,%d)" % (lo, hi), sel)
    run("1000 x (0,1)", [(O.MODE_AND, [0, 1])] * 1000)
    run("1000 x (0,255)", [(O.MODE_AND, [0, 255])] * 1000)
    run("1000 x (20,250)", [(O.MODE_AND, [20, 250])] * 1000)
    run("1000 x (100,200)", [(O.MODE_AND, [100, 200])] * 1000)
    dev.close()


if __name__ == "__main__":
    main()
