#!/usr/bin/env python
"""CPU model (no GPU): how many doc-matrix gathers the shared-union launch's stage A would keep if a doc had to
pass "leader score (its own fieldnorm byte) + weights of the later lists >= threshold" BEFORE its 8-byte doc-matrix
word is gathered.  Final thresholds (the k-th best score of the exhaustive run): an optimistic floor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 5
thr_scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
seg = O.synth_segment(10_000_000, n_terms=256)
qs = O.zipf_queries(1000, nt, 256, seed=20260922)[:nq]
table = np.array(O.fieldnorm_table(), dtype=np.float64)
avg = seg.avg_fieldnorm
cache = (1.2 * (1 - 0.75 + 0.75 * table / avg)).astype(np.float32)
dec = {}
def plist(t):
    if t not in dec:
        d, f = O.decode_postings(seg, int(t))
        tfn = f.astype(np.float32) / (f.astype(np.float32) + cache[seg.fieldnorm[d]])
        dec[t] = tfn
    return dec[t]
tot_blocks = tot_docs = tot_pass = 0
by_pos = np.zeros((nt, 3))
for q in qs:
    ws = O.default_weights(seg, q.tolist(), O.MODE_OR)
    w = np.array([x.weight for x in ws], dtype=np.float64)
    order = np.argsort(-w, kind="stable")
    terms = q[order]; w = w[order]
    hits = O.search(seg, q.tolist(), O.MODE_OR, k, pruned=False)
    thr = hits[-1][0] * thr_scale if len(hits) >= k else 0.0
    for i in range(nt):
        suffix = w[i:].sum()
        if suffix < thr: continue
        tfn = plist(int(terms[i]))
        n = len(tfn); nb = (n + 127) // 128
        pad = np.zeros(nb * 128, np.float32); pad[:n] = tfn
        bm = pad.reshape(nb, 128).max(axis=1)
        rest = suffix - w[i]
        wanted = w[i] * bm + rest >= thr
        docs_w = int(wanted.sum()) * 128
        ps = (w[i] * pad.reshape(nb, 128)[wanted] + rest >= thr).sum()
        tot_blocks += int(wanted.sum()); tot_docs += docs_w; tot_pass += int(ps)
        by_pos[i] += (int(wanted.sum()), docs_w, int(ps))
print("queries", nq, "k", k, "blocks wanted", tot_blocks, "docs in them", tot_docs, "docs passing the own-norm bound", tot_pass,
      "ratio %.3f" % (tot_pass / max(1, tot_docs)))
for i in range(nt):
    print(" position", i, "blocks %d docs %d pass %d (%.3f)" % (by_pos[i][0], by_pos[i][1], by_pos[i][2], by_pos[i][2] / max(1, by_pos[i][1])))
