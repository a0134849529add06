#!/usr/bin/env python
"""Boolean queries through the shared launch (tq_ashare.hip, boolean leads) against the per-query
union kernel (TQ_BSHARE=0) on the same batch: the bench's four boolean shapes over the PROBE_VOCAB most
frequent terms (every list has a bitmap: every query is eligible).  Knobs come from the environment
(one process per setting); TQ_DEBUG counters as tools/probe_ashare.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd as T  # noqa: E402

vocab = int(os.environ.get("PROBE_VOCAB", "48"))
nq = int(os.environ.get("PROBE_QUERIES", "2000"))
ratio = int(os.environ.get("PROBE_DENSE_RATIO", "0"))
seg = O.synth_segment(10_000_000, n_terms=256)
dev = T.DeviceIndex([seg], devices=[0])
dev.set_option("timing", 1)
if ratio:
    dev.set_option("dense_ratio", ratio)
if os.environ.get("PROBE_DENSE_BUDGET"):
    dev.set_option("dense_budget_x", int(os.environ["PROBE_DENSE_BUDGET"]))
ids = O.zipf_queries(nq, 4, vocab, seed=20260924)
M, S, N = T.MUST, T.SHOULD, T.MUST_NOT
shapes = [(3, [M, M, M], [0, 1, 1]), (4, [M, M, M, M], [0, 0, 1, 1]), (3, [M, S, N], None), (3, [M, M, M], [0, 0, 1])]
only = os.environ.get("PROBE_SHAPE")
qs = []
for i, q in enumerate(ids):
    nt, occ, cof = shapes[int(only) if only else i % len(shapes)]
    qs.append((T.MODE_BOOL, q.tolist()[:nt], occ, cof, 0))
dev.set_option("exhaustive", 0)
dev.prepare(qs)
for _ in range(3):
    dev.search_prepared(10)
t0 = time.perf_counter()
for _ in range(10):
    dev.search_prepared(10)
wall = (time.perf_counter() - t0) / 10
st = dev.last_batch_stats()
print("%s kernel %.3f ms host %.3f ms wall/step %.3f ms counter %.4g tasks %d kernels %s" %
      (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith(("TQ_", "PROBE_"))),
       st["kernel_ms"], st["host_plan_ms"], wall * 1e3, st["matches"], st["chunks"], st["kernels"]))
ss = dev.segment_stats(0)
print("segment: terms %d dense lists %d columns %d bitmap MB %.0f docmat MB %.0f" % (ss["n_terms"], ss["n_dense_lists"], ss["n_docmat_columns"], ss["bitmap_bytes"] / 1e6, ss["docmat_bytes"] / 1e6))
dev.close()
