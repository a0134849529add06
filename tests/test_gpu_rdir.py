"""Range directories (round 6): the probe structure of lists below "dense_ratio" in the shared
intersection launch — one u32 per posting in posting order + a directory of posting counts per 2^S docs
(rdir_lookup in tantivy_amd/csrc/tq_common.hpp, built by tq_term_prepare / tq_term_prepare_batch) where a
max_doc / 4-byte bitmap + rank directory + tf bytes from the probe pool stood before.

A crafted segment holds the cases the structure distinguishes: lists at the minimum length, lists clustered into a
few ranges (a range with dozens of entries: the lookup walks), lists spread evenly, a tf beyond the entry's 16 bits
(the escape to the packed value), a list too short for a directory, dense lists next to them.  Every pair of lists
as a 2-term intersection, pruned == exhaustive and against the oracle (block_wand_intersection.rs /
intersection.rs), through both build paths (terms prepared one by one and a batch's new terms together), and with
"rdir_budget_x" = 0 (the probe pool's bitmaps again): the same rows."""
import itertools

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

MAX_DOC = 400_000


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def _lists(rng):
    def spread(df, max_tf=6):
        docs = np.sort(rng.choice(MAX_DOC, size=df, replace=False))
        return list(zip(docs.tolist(), rng.integers(1, max_tf + 1, size=df).tolist()))

    def clustered(df, lo, width):
        docs = np.sort(rng.choice(width, size=df, replace=False)) + lo
        return list(zip(docs.tolist(), rng.integers(1, 5, size=df).tolist()))

    lists = []
    lists += [spread(df) for df in (256, 257, 300, 511, 1000, 1283, 5000, 12_000, 24_000)]  # directories down to the dense threshold (max_doc / 128)
    lists += [clustered(700, 100_000, 900), clustered(3000, 250_000, 4000), clustered(260, 399_000, 1000)]
    lists += [spread(255), spread(40)]                                                      # too short for a directory
    lists += [spread(60_000), spread(120_000), spread(200_000)]                             # dense: bitmaps of their own
    big = spread(2000)
    big[17] = (big[17][0], 70_000)   # beyond the entry's 16 bits
    big[900] = (big[900][0], 65_535)  # exactly the escape value
    big[901] = (big[901][0], 65_534)
    lists.append(big)
    # companions that share docs with the sparse lists (a random pair of sparse lists intersects in nothing)
    base = lists[4]
    lists.append(sorted(set(d for d, _ in base[::2]) | set(d for d, _ in lists[9][::3]) | set(d for d, _ in big[::2])))
    lists[-1] = [(d, int(rng.integers(1, 4))) for d in lists[-1]]
    return lists


def _rows(out, qi):
    sc, _, dc, ct = out
    return [(float(sc[qi, j]), int(dc[qi, j])) for j in range(int(ct[qi]))]


def _run(ta, seg, queries, k, one_by_one, rdir_budget):
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("ashare_min_batch", 1)
        if rdir_budget is not None:
            dev.set_option("rdir_budget_x", rdir_budget)
        if one_by_one:  # every query but the first names ONE new term: tq_term_prepare's path builds its directory
            n = len(seg.terms)
            for t in range(n - 1):
                dev.search([(O.MODE_AND, [t, t + 1])], k)
        dev.set_option("exhaustive", 0)
        pr = dev.search(queries, k)
        st = dev.last_batch_stats()
        dev.set_option("exhaustive", 1)
        ex = dev.search(queries, k)
        stats = dev.segment_stats(0)
    finally:
        dev.close()
    for a, b in zip(pr, ex):
        assert np.array_equal(a, b)
    return pr, st, stats


def test_range_directories_against_the_oracle_and_the_probe_pool(ta):
    rng = np.random.default_rng(606)
    lists = _lists(rng)
    fieldnorms = rng.integers(1, 60, size=MAX_DOC).tolist()
    seg = O.build_segment(MAX_DOC, lists, fieldnorms=fieldnorms)
    n = len(lists)
    queries = [(O.MODE_AND, [a, b]) for a, b in itertools.permutations(range(n), 2)]
    k = 10
    ref = None
    table_bytes = {}
    for one_by_one, budget in ((False, None), (True, None), (False, 0)):
        pr, st, stats = _run(ta, seg, queries, k, one_by_one, budget)
        table_bytes[(one_by_one, budget)] = stats["term_table_bytes"]
        assert st["kernel_mask"] & ta.binding.KERNEL_ASHARE, st
        if budget is None:
            # directories, not bitmaps: the bitmap bytes are the dense lists' own (+ their tf bytes, range maxima, ...)
            assert stats["bitmap_bytes"] < (stats["n_dense_lists"] + 1) * (MAX_DOC // 4 + MAX_DOC // 2), stats
        else:
            assert stats["bitmap_bytes"] > (stats["n_dense_lists"] + 8) * (MAX_DOC // 4), stats
        if ref is None:
            ref = pr
            for qi, q in enumerate(queries):
                want = O.search(seg, q[1], q[0], k, pruned=False)
                got = _rows(pr, qi)
                assert got == [(float(np.float32(s)), d) for s, d in want], (q, got[:3], want[:3])
        else:
            for a, b in zip(pr, ref):
                assert np.array_equal(a, b), (one_by_one, budget)
    # the directories are counted with the terms' tables: at least 4 bytes per posting of the lists that got one
    n_entries = sum(len(l) for l in lists if 256 <= len(l) < MAX_DOC // 128)  # (dense_ratio: 128)
    for key in ((False, None), (True, None)):
        assert table_bytes[key] >= table_bytes[(False, 0)] + 4 * n_entries, table_bytes


def test_shared_unions_probe_sparse_lists_through_their_directories(ta):
    """The shared-union launch (tq_ushare.hip) scores a candidate's later lists through bitmap word -> tf byte; a list
    without a bitmap was seeked and block-searched, and is now asked through its range directory.  3- and 5-term unions
    over the crafted segment's lists, pruned == exhaustive, against the oracle (block_wand_union.rs / union.rs), and the
    same rows with "rdir_budget_x" = 0 (the block search)."""
    rng = np.random.default_rng(607)
    lists = _lists(rng)
    fieldnorms = rng.integers(1, 60, size=MAX_DOC).tolist()
    seg = O.build_segment(MAX_DOC, lists, fieldnorms=fieldnorms)
    n = len(lists)
    queries = []
    for i in range(600):
        nt = 3 if i % 2 else 5
        queries.append((O.MODE_OR, rng.permutation(n)[:nt].tolist()))
    k = 10
    got = {}
    for budget in (None, 0):
        dev = ta.DeviceIndex([seg])
        try:
            if budget is not None:
                dev.set_option("rdir_budget_x", budget)
            dev.set_option("timing", 1)
            dev.set_option("exhaustive", 0)
            pr = dev.search(queries, k)
            st = dev.last_batch_stats()
            assert st["kernel_mask"] & ta.binding.KERNEL_USHARE, st
            dev.set_option("exhaustive", 1)
            ex = dev.search(queries, k)
        finally:
            dev.close()
        assert np.array_equal(pr[2], ex[2]) and np.array_equal(pr[3], ex[3])
        assert np.allclose(pr[0], ex[0], rtol=1e-5, atol=0)
        got[budget] = pr
    assert np.array_equal(got[None][2], got[0][2]) and np.array_equal(got[None][0], got[0][0])
    for qi in range(0, len(queries), 7):
        q = queries[qi]
        want = O.search(seg, q[1], q[0], k, pruned=False)
        rows = _rows(got[None], qi)
        assert sorted(d for _, d in rows) == sorted(d for _, d in want), (q, rows, want)
        for (a, _), (b, _) in zip(rows, want):
            assert abs(a - b) <= 1e-5 * max(abs(a), abs(b)), (q, rows, want)
