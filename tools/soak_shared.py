#!/usr/bin/env python
"""Soak run of the batched launches (ashare / bshare / ushare / tree / phrase sweep and the per-query kernels next to them):
SEEDS random configurations — segment size, vocabulary, which lists get bitmaps / probe slots / range directories, k,
batch size — each a random mixed batch (2..4-term intersections, 2..6-term unions, the bench's boolean shapes, phrases), checked the way tests/test_gpu_round5.py checks the bench's streams: pruned == exhaustive on EVERY
query (docs and counts bit for bit, scores bit for bit for two lists), and a sample of every kernel family of the batch
against the oracle.  One line per seed; exit code 1 on the first mismatch.
    SEEDS=40 FIRST=0 python tools/soak_shared.py        (DOCS=3000000,8000000: other segment sizes)"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd as ta  # noqa: E402
from tests import test_gpu_round5 as R5  # noqa: E402


def oracle_rows(seg, q, k):
    if q[0] in (O.MODE_AND, O.MODE_OR, O.MODE_PHRASE):
        return O.search(seg, q[1], q[0], k, pruned=False)
    return O.bool_search(seg, q[1], q[2], k, q[3], q[4])


def check_batch(dev, seg, queries, k, per_family):
    """tests/test_gpu_round5.py's _check_batch with phrases, and with near-ties across the k-th rank allowed for sums
    of three or more terms (their order is not canonical in the reference either): scores rank by rank within 1e-5,
    docs equal except among those within 1e-5 of the k-th score."""
    dev.set_option("timing", 1)
    dev.set_option("record_query_kernels", 1)
    dev.set_option("exhaustive", 0)
    pr = dev.search(queries, k)
    st = dev.last_batch_stats()
    kern = dev.last_batch_query_kernels(len(queries))
    dev.set_option("record_query_kernels", 0)
    dev.set_option("exhaustive", 1)
    ex = dev.search(queries, k)
    assert np.array_equal(pr[3], ex[3]), "counts differ between pruned and exhaustive"
    for qi, q in enumerate(queries):
        two = q[0] == O.MODE_AND and len(q[1]) == 2
        if two:
            assert np.array_equal(pr[2][qi], ex[2][qi]) and np.array_equal(pr[0][qi], ex[0][qi]), ("pruned != exhaustive", q)
        else:
            assert np.allclose(pr[0][qi], ex[0][qi], rtol=1e-5, atol=0), ("pruned != exhaustive (scores)", q, pr[0][qi], ex[0][qi])
            if not np.array_equal(pr[2][qi], ex[2][qi]):
                n = int(pr[3][qi])
                kth = float(pr[0][qi, n - 1])
                for a in set(pr[2][qi, :n].tolist()) ^ set(ex[2][qi, :n].tolist()):
                    sc = [float(x[0][qi, j]) for x in (pr, ex) for j in range(n) if int(x[2][qi, j]) == a][0]
                    assert R5.rel_close(sc, kth, 1e-5), ("pruned != exhaustive (docs)", q, pr[2][qi], ex[2][qi])
    sample = R5._sample_by_family(kern, per_family)
    for qi in sample:
        q = queries[qi]
        want = oracle_rows(seg, q, k)
        got = [(float(pr[0][qi, j]), int(pr[2][qi, j])) for j in range(int(pr[3][qi]))]
        assert len(got) == len(want), (q, got, want)
        if q[0] == O.MODE_AND and len(q[1]) == 2:
            assert got == [(float(np.float32(s)), d) for s, d in want], (q, got[:4], want[:4])
            continue
        assert all(R5.rel_close(a, b, 1e-5) for (a, _), (b, _) in zip(got, want)), (q, got, want)
        if got:
            kth = got[-1][0]
            for d in set(d for _, d in got) ^ set(d for _, d in want):
                sc = [s_ for s_, dd in got + want if dd == d][0]
                assert R5.rel_close(sc, kth, 1e-5), (q, got, want)
    return st, kern, len(sample)


def batch(rng, n, vocab, with_phrases, seed):
    M, S, N = ta.MUST, ta.SHOULD, ta.MUST_NOT
    shapes = [(3, [M, M, M], [0, 1, 1], 0), (4, [M, M, M, M], [0, 0, 1, 1], 0), (3, [M, S, N], None, 0), (3, [M, M, M], [0, 0, 1], 0),
              (3, [M, S, S], None, 1), (4, [S, S, S, N], None, 2), (2, [M, N], None, 0), (4, [M, M, S, S], None, 0)]
    ranks = {nt: O.zipf_queries(n, nt, vocab, seed=seed * 7 + nt) for nt in (2, 3, 4, 5, 6)}
    kinds = rng.choice(5 if with_phrases else 4, size=n, p=None)
    qs = []
    for i in range(n):
        kind = int(kinds[i])
        if kind == 0:
            nt = int(rng.choice([2, 2, 2, 3, 4]))
            qs.append((O.MODE_AND, ranks[nt][i].tolist()))
        elif kind == 1:
            nt = int(rng.choice([2, 3, 5, 5, 6]))
            qs.append((O.MODE_OR, ranks[nt][i].tolist()))
        elif kind in (2, 3):
            nt, occ, cof, msm = shapes[int(rng.integers(len(shapes)))]
            qs.append((ta.MODE_BOOL, ranks[4][i].tolist()[:nt], occ, cof, msm))
        else:
            a = int(rng.integers(0, min(vocab, 24) - 3))
            qs.append((O.MODE_PHRASE, [a, a + 1, a + 2][: int(rng.choice([2, 3]))]))
    return qs


def main():
    n_seeds = int(os.environ.get("SEEDS", "20"))
    first = int(os.environ.get("FIRST", "0"))
    bad = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(9000 + seed)
        sizes = [int(x) for x in os.environ["DOCS"].split(",")] if os.environ.get("DOCS") else [150_000, 300_000, 700_000, 1_500_000]
        n_docs = int(rng.choice(sizes))
        vocab = int(rng.choice([48, 256, 1024, 4096]))
        with_pos = vocab <= 256 and n_docs <= 700_000
        k = int(rng.choice([1, 10, 10, 100]))
        n = int(rng.choice([40, 300, 1200, 3000]))
        opts = {"dense_ratio": int(rng.choice([8, 64, 64, 512])), "probe_budget_x": int(rng.choice([0, 2, 16])),
                "rdir_budget_x": int(rng.choice([0, 4, 4])), "ashare_min_batch": int(rng.choice([1, 16, 16, 512]))}
        t0 = time.time()
        seg = O.synth_segment(n_docs, n_terms=vocab, with_positions=with_pos, phrase_terms=24)
        qs = batch(rng, n, vocab, with_pos, seed)
        dev = ta.DeviceIndex([seg])
        try:
            for name, v in opts.items():
                dev.set_option(name, v)
            st, kern, n_checked = check_batch(dev, seg, qs, k, 32)
            # a second, different batch on the same (now warm) segment: tables exist, slots may be evicted
            qs2 = batch(rng, n, vocab, with_pos, seed + 100000)
            st2, kern2, n2 = check_batch(dev, seg, qs2, k, 32)
            print("seed %d ok: %d docs, %d terms, k %d, %d queries, %s -> kernels %s | %s, %d + %d oracle-checked, %.1f s" %
                  (seed, n_docs, vocab, k, n, opts, "+".join(st["kernels"]), "+".join(st2["kernels"]), n_checked, n2, time.time() - t0),
                  flush=True)
        except Exception:
            bad += 1
            print("seed %d FAILED: %d docs, %d terms, k %d, %d queries, %s" % (seed, n_docs, vocab, k, n, opts), flush=True)
            traceback.print_exc(limit=4)
            sys.stdout.flush()
            if os.environ.get("STOP", "1") != "0":
                break
        finally:
            dev.close()
    print("soak: %d seeds, %d failed" % (n_seeds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
