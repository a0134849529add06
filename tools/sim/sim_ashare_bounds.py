#!/usr/bin/env python
"""CPU model of the shared-intersection launch's candidate counts under different bounds on the
non-leader list (no GPU): for the headline batch (10 000 Zipf 2-term ANDs, 10M docs) and each
distinct query, how many docs of the leader would reach the scoring stage (stage C of
tq_ashare.hip) at the query's FINAL threshold when the secondary list is bounded by

  weight        rest = w1                                  (round 4: Bm25Weight::max_score analogue)
  range R       rest = w1 * max tf/(tf+norm) of list 1 over the doc's R-doc range   (block-max analogue,
                block_wand_intersection.rs:59-85 with fixed doc ranges instead of 128-posting blocks)
  rtf R         rest = w1 * tfmax_R/(tfmax_R + norm(doc))  (max tf of the range, the doc's own norm)
  class         the doc's tf in list 1 known exactly when it is 1 or 2 (2 bits per doc and column),
                else the list's max tf with the doc's own norm

Thresholds rise during a real launch, so absolute counts are a floor; the ratios between the
schemes are what the design decision needs.  Usage: python tools/sim_ashare_bounds.py [n_queries]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    N = 10_000_000
    seg = O.synth_segment(N, n_terms=256)
    fn = np.frombuffer(seg.fieldnorm, dtype=np.uint8)
    table = np.array(O.fieldnorm_table(), dtype=np.float32)
    avg = np.float32(seg.total_num_tokens) / np.float32(N)
    cache = (np.float32(1.2) * (np.float32(0.25) + np.float32(0.75) * table / avg)).astype(np.float32)
    norm_doc = cache[fn]  # per doc
    t0 = time.time()
    docs, tfa = [], []
    for t in range(256):
        d, tf = O.decode_postings(seg, t)
        docs.append(d.astype(np.int64))
        a = np.zeros(N, dtype=np.uint8)
        a[d] = np.minimum(tf, 255)
        tfa.append(a)
    print("decoded in %.1f s" % (time.time() - t0), flush=True)
    # per-list range maxima
    RANGES = (256, 1024, 4096)
    rmax_tfn = {R: [] for R in RANGES}
    rmax_tf = {R: [] for R in RANGES}
    for t in range(256):
        a = tfa[t].astype(np.float32)
        tfn = np.where(a > 0, a / (a + norm_doc), 0).astype(np.float32)
        for R in RANGES:
            nr = (N + R - 1) // R
            pad = nr * R - N
            x = np.pad(tfn, (0, pad)).reshape(nr, R).max(axis=1)
            rmax_tfn[R].append(x)
            y = np.pad(tfa[t], (0, pad)).reshape(nr, R).max(axis=1)
            rmax_tf[R].append(y)
    list_max_tf = [int(tfa[t].max()) for t in range(256)]
    print("range tables in %.1f s" % (time.time() - t0), flush=True)
    dfs = [len(d) for d in docs]
    qs = O.zipf_queries(n_q, 2, 256, seed=20260921)
    seen = {}
    for q in qs:
        a, b = int(q[0]), int(q[1])
        key = (a, b) if dfs[a] <= dfs[b] else (b, a)  # leader (rarer) first
        seen[key] = seen.get(key, 0) + 1
    print("%d queries, %d distinct" % (len(qs), len(seen)))
    k = 10
    tot = {"matches": 0, "weight": 0, "class": 0, "collected_floor": 0, "weight_x_groups": 0, "range1024_x_groups": 0, "leader_docs_x_groups": 0, "leader_pass_weight_x_groups": 0}
    for R in RANGES:
        tot["range%d" % R] = 0
        tot["rtf%d" % R] = 0
    tot["min(range1024,rtf1024)"] = 0
    tot["class+range1024"] = 0
    for (l, s), mult in seen.items():
        w = O.default_weights(seg, [l, s], O.MODE_AND)
        w0, w1 = np.float32(w[0].weight), np.float32(w[1].weight)
        d0 = docs[l]
        tf0 = tfa[l][d0].astype(np.float32)
        tf1 = tfa[s][d0]
        m = tf1 > 0
        d = d0[m]
        if len(d) == 0:
            continue
        n0 = norm_doc[d]
        f0 = tf0[m]
        f1 = tf1[m].astype(np.float32)
        s0 = w0 * (f0 / (f0 + n0))
        s1 = w1 * (f1 / (f1 + n0))
        sc = s0 + s1
        thr = np.partition(sc, -k)[-k] if len(sc) >= k else np.float32(0)
        tot["matches"] += len(d)
        # the threshold as the kernel knows it: k-th largest of 64 hashed slots, upper 16 bits of the score
        thr16 = (np.array([thr], dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)[0]
        tot["weight_thr16"] = tot.get("weight_thr16", 0) + int((s0 + w1 >= thr16).sum())
        thr24 = (np.array([thr], dtype=np.float32).view(np.uint32) & np.uint32(0xFFFFFF00)).view(np.float32)[0]
        tot["weight_thr24"] = tot.get("weight_thr24", 0) + int((s0 + w1 >= thr24).sum())
        tot["range1024_thr16"] = tot.get("range1024_thr16", 0) + int((s0 + w1 * rmax_tfn[1024][s][d // 1024] >= thr16).sum())
        tot["range1024_thr24"] = tot.get("range1024_thr24", 0) + int((s0 + w1 * rmax_tfn[1024][s][d // 1024] >= thr24).sum())
        tot["collected_floor"] += int((sc >= thr).sum())
        tot["weight"] += int((s0 + w1 >= thr).sum())
        groups = (mult + 31) // 32  # a family is evaluated once per group of <= 32 leads
        tot["weight_x_groups"] += groups * int((s0 + w1 >= thr).sum())
        tot["leader_docs_x_groups"] += groups * len(d0)
        tot["leader_pass_weight_x_groups"] += groups * int((w0 * (tf0 / (tf0 + norm_doc[d0])) + w1 >= thr).sum())
        br = {}
        for R in RANGES:
            b_r = w1 * rmax_tfn[R][s][d // R]
            tm = rmax_tf[R][s][d // R].astype(np.float32)
            b_t = w1 * (tm / (tm + n0))
            br[R] = (b_r, b_t)
            tot["range%d" % R] += int((s0 + b_r >= thr).sum())
            if R == 1024:
                tot["range1024_x_groups"] += groups * int((s0 + b_r >= thr).sum())
            tot["rtf%d" % R] += int((s0 + b_t >= thr).sum())
        tot["min(range1024,rtf1024)"] += int((s0 + np.minimum(br[1024][0], br[1024][1]) >= thr).sum())
        lm = np.float32(list_max_tf[s])
        b_c = np.where(f1 <= 2, s1, w1 * (lm / (lm + n0)))
        tot["class"] += int((s0 + b_c >= thr).sum())
        b_c2 = np.where(f1 <= 2, s1, np.minimum(br[1024][0], br[1024][1]))
        tot["class+range1024"] += int((s0 + b_c2 >= thr).sum())
    print("distinct-query totals (each distinct query once), final thresholds:")
    for key, v in tot.items():
        print("  %-26s %12d" % (key, v))


if __name__ == "__main__":
    main()
