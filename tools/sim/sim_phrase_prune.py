#!/usr/bin/env python
"""CPU model: how many phrase candidates (docs holding all the phrase's terms) could still enter the top-k if a candidate
is bounded by bm25(sum of idfs, norm, min tf of its terms) (phrase count <= min tf) against the FINAL k-th score."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
docs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
seg = O.synth_segment(docs, n_terms=256, with_positions=True, phrase_terms=64)
qs, k = bench.build_queries(O, "phrase3", 1000, None)
table = np.array(O.fieldnorm_table(), dtype=np.float64)
cache = (1.2 * (1 - 0.75 + 0.75 * table / seg.avg_fieldnorm)).astype(np.float32)
dec = {}
def plist(t):
    if t not in dec: dec[t] = O.decode_postings(seg, t)
    return dec[t]
tot_c = tot_m = tot_p = 0
for q in qs[:n]:
    hits = O.search(seg, q[1], O.MODE_PHRASE, k, pruned=False, phrase_offsets=q[2])
    thr = hits[-1][0] if len(hits) >= k else 0.0
    w = O.default_weights(seg, q[1], O.MODE_PHRASE)[0].weight
    d0, f0 = plist(q[1][0]); mt = None
    cur = d0; tfs = [f0]
    for t in q[1][1:]:
        d, f = plist(t)
        both, ia, ib = np.intersect1d(cur, d, assume_unique=True, return_indices=True)
        tfs = [x[ia] for x in tfs] + [f[ib]]
        cur = both
    mintf = np.minimum.reduce(tfs).astype(np.float32)
    norm = cache[seg.fieldnorm[cur]]
    ub = np.float32(w) * (mintf / (mintf + norm))
    dm, sm = O.match_all(seg, q[1], O.MODE_PHRASE, phrase_offsets=q[2])
    tot_c += len(cur); tot_m += len(dm); tot_p += int((ub * 1.000002 >= thr).sum())
print("queries", n, "candidates", tot_c, "matches", tot_m, "candidates whose bound reaches the final k-th score", tot_p, "%.3f" % (tot_p / max(1, tot_c)))
