#!/usr/bin/env python
"""One query per call: and_kernel's wave time by region (the kernel's own phase timers, option "debug" = region << 16:
1 set-up + flush, 2 threshold + pre-filter, 3 stage A, 4 stage B, 5 stage C), for a few queries of the headline
stream, next to the call's wall time.  Usage (GPU box): python tools/r5_single_phases.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
ids = O.zipf_queries(64, 2, 256, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
dev.set_option("timing", 1)
dev.search(qs, 10)
B = int(os.environ.get("BATCH", "1"))
for qi in ((0, 1, 2, 3, 5, 8) if B == 1 else (0, 16, 32)):
    q = qs[qi:qi + B]
    dfs = sorted(seg.terms[t].doc_freq for t in q[0][1])
    dev.prepare(q)
    row = []
    for ph in (0, 1, 2, 3, 4, 5):
        dev.set_option("debug", ph << 16)
        t = []
        for _ in range(12):
            t1 = time.perf_counter()
            dev.search_prepared(10)
            t.append(time.perf_counter() - t1)
        st = dev.last_batch_stats()
        row.append("%s=%d" % ("wall_us" if ph == 0 else "r%d" % ph, sorted(t)[6] * 1e6 if ph == 0 else st["matches"]))
        if ph == 0:
            row.append("kernel_us=%d tiles=%d" % (st["kernel_ms"] * 1e3, st.get("tiles", 0)))
    print("%d quer%s from %s leader %d blocks, other df %d: %s kernels %s" % (B, "y" if B == 1 else "ies", q[0][1], (dfs[0] + 127) // 128, dfs[1], " ".join(row), "+".join(st["kernels"])), flush=True)
dev.set_option("debug", 0)
dev.close()
