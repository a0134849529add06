"""GPU parity tests added in round 3: BASELINE config 5's shape (several segments under global BM25
statistics, mixed AND / OR stream, merge_top_k across segments) against the ORACLE at sizes of its
own — 8 x 200k docs and 2 x 10M docs, with deletes on one segment — and the new introspection."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rel_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def _alive_bytes(max_doc, deleted):
    """BitSet::serialize (common/src/bitset.rs:215-223) of the alive set."""
    bits = np.ones(((max_doc + 63) // 64) * 64, dtype=np.uint8)
    bits[max_doc:] = 0
    bits[np.asarray(sorted(deleted), dtype=np.int64)] = 0
    return np.uint32(max_doc).tobytes() + np.packbits(bits, bitorder="little").tobytes()


def _mixed_stream(n, n_terms, seed):
    a = O.zipf_queries(n // 2, 2, n_terms, seed=seed)
    o = O.zipf_queries(n - n // 2, 5, n_terms, seed=seed + 1)
    return [(O.MODE_AND, a[i // 2].tolist()) if i % 2 == 0 else (O.MODE_OR, o[i // 2].tolist())
            for i in range(n)]


def _oracle_merged(segs, deleted, queries, k):
    """Searcher::search restated with the oracle: per segment every match under the index-wide
    Bm25Weights (bm25.rs:27-50), deleted docs dropped (sort_by_score.rs:44-53), the segment's
    top-k by (score desc, doc asc), then merge_top_k (sort_key_top_collector.rs:76-95)."""
    nd = sum(s.max_doc for s in segs)
    nt = sum(s.total_num_tokens for s in segs)
    out = []
    for mode, terms in queries:
        dfs = [sum(s.terms[t].doc_freq for s in segs) for t in terms]
        hits = []
        for o, s in enumerate(segs):
            w = O.default_weights(s, terms, mode, total_num_docs=nd, total_num_tokens=nt, dfs=dfs)
            docs, scores = O.match_all(s, terms, mode, weights=w)
            if deleted.get(o) is not None and len(docs):
                keep = ~np.isin(docs, deleted[o])
                docs, scores = docs[keep], scores[keep]
            order = np.lexsort((docs, -scores.astype(np.float64)))[:k]
            hits += [(float(scores[i]), o, int(docs[i])) for i in order]
        out.append(O.merge_top_k(hits, 0, k))
    return out


def _check(got, want, queries):
    sc, ords, docs, cnt = got
    for qi, w in enumerate(want):
        g = [(float(sc[qi, j]), int(ords[qi, j]), int(docs[qi, j])) for j in range(int(cnt[qi]))]
        assert len(g) == len(w), (queries[qi], g, w)
        for a, b in zip(g, w):
            assert a[1:] == b[1:], (queries[qi], g, w)
            assert rel_close(a[0], b[0], 1e-5), (queries[qi], g, w)


def test_eight_segments_mixed_stream_against_the_oracle(ta):
    """8 x 200k docs (seeds differ per segment), 120 queries of the mixed stream, k = 10, a third
    of segment 3 deleted: pruned and exhaustive device results == the oracle's, through
    Searcher::search and through the ShardRunner (collect_segment x 8 -> all-gather (one-rank RCCL)
    -> merge_top_k on one stream)."""
    from tantivy_amd import distributed as D

    segs = [O.synth_segment(200_000, n_terms=64, segment_ord=o) for o in range(8)]
    rng = np.random.default_rng(5)
    dele = np.sort(rng.choice(segs[3].max_doc, size=segs[3].max_doc // 3, replace=False))
    queries = _mixed_stream(120, 64, 777)
    want = _oracle_merged(segs, {3: dele}, queries, 10)
    dev = ta.DeviceIndex(segs, devices=[0])
    try:
        dev.set_alive_bitset(_alive_bytes(segs[3].max_doc, dele.tolist()), 3)
        for ex in (0, 1):
            dev.set_option("exhaustive", ex)
            _check(dev.search(queries, 10), want, queries)
    finally:
        dev.close()
    run = D.ShardRunner(segs, 0, 0, 1, force_exchange=True)
    comm = D.Comm(run.dev.ctx, 0, 0, 1, lambda raw: raw)
    try:
        run.comm = comm
        run.dev.set_alive_bitset(_alive_bytes(segs[3].max_doc, dele.tolist()), 3)
        run.set_option("timing", 1)
        run.prepare(queries, 10)
        for _ in range(2):
            run.enqueue()
        run.synchronize()
        _check(run.results(), want, queries)
        st = run.batch_stats()
        assert st["host_plan_ms"] > 0.0 and st["kernel_ms"] > 0.0
        assert run.exchange_ms() > 0.0
    finally:
        comm.close()
        run.close()


def test_two_full_size_segments_mixed_stream_against_the_oracle(ta):
    """2 x 10M docs, 24 queries of the mixed stream (heavy and rare terms), k = 10, deletes on
    segment 1, global statistics: pruned == exhaustive == oracle."""
    segs = [O.synth_segment(10_000_000, n_terms=256, segment_ord=o) for o in range(2)]
    rng = np.random.default_rng(9)
    dele = np.sort(rng.choice(segs[1].max_doc, size=segs[1].max_doc // 5, replace=False))
    queries = _mixed_stream(20, 256, 4242) + [(O.MODE_AND, [0, 1]), (O.MODE_OR, [0, 1, 2, 3, 4]),
                                               (O.MODE_AND, [3, 250]), (O.MODE_OR, [200, 7, 90, 1, 255])]
    want = _oracle_merged(segs, {1: dele}, queries, 10)
    dev = ta.DeviceIndex(segs, devices=[0])
    try:
        dev.set_alive_bitset(_alive_bytes(segs[1].max_doc, dele.tolist()), 1)
        for ex in (0, 1):
            dev.set_option("exhaustive", ex)
            _check(dev.search(queries, 10), want, queries)
    finally:
        dev.close()


def test_default_mode_is_the_references_pruned_execution(ta):
    """The raw library default is block-max pruning (what the reference executes); a caller that
    never touches "exhaustive" gets the same top-k and far fewer docs scored."""
    seg = O.synth_segment(2_000_000, n_terms=64)
    qs = [(O.MODE_AND, q.tolist()) for q in O.zipf_queries(200, 2, 64, seed=31)]
    dev = ta.DeviceIndex([seg])
    try:
        a = dev.search(qs, 10)
        scored_default = dev.last_batch_stats()["matches"]
        dev.set_option("exhaustive", 1)
        b = dev.search(qs, 10)
        scored_all = dev.last_batch_stats()["matches"]
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert scored_default < scored_all // 4, (scored_default, scored_all)
    finally:
        dev.close()


def test_segment_stats_account_for_the_resident_bytes(ta):
    """tq_segment_get_stats: tantivy's bytes are the uploaded sub-files; the derived tables are
    reported by kind and stay within the dense budget."""
    seg = O.synth_segment(400_000, n_terms=48, with_positions=True, phrase_terms=8)
    dev = ta.DeviceIndex([seg])
    try:
        dev.search([(O.MODE_AND, [0, 1]), (O.MODE_OR, [2, 3, 40]), (O.MODE_PHRASE, [0, 1, 2])], 10)
        st = dev.segment_stats(0)
        assert st["index_bytes"] == seg.idx_len
        assert st["positions_bytes"] == seg.pos_len
        assert st["fieldnorm_bytes"] == seg.max_doc
        assert st["n_terms"] >= 5 and st["term_table_bytes"] > 0  # (terms are prepared on first use)
        assert st["n_dense_lists"] >= 1 and st["bitmap_bytes"] >= st["n_dense_lists"] * (seg.max_doc // 4)
        assert st["docmat_bytes"] in (0, 8 * seg.max_doc, 16 * seg.max_doc)  # (doc matrix, + the tf-class matrix since round 6)
        assert st["bitmap_bytes"] + st["docmat_bytes"] + st["posdir_bytes"] <= st["dense_budget_bytes"]
        assert st["derived_bytes"] == (st["term_table_bytes"] + st["bitmap_bytes"] + st["docmat_bytes"]
                                       + st["posdir_bytes"])
    finally:
        dev.close()


def _big_tf_segment(with_positions):
    """20k docs, 8 lists (5 dense with bitmaps / tf bytes, 3 sparse), term freqs mostly 1..3 with
    255, 256, 300 and 1000 planted in every list — the saturated value of the byte-wide tfs and
    the values beyond it."""
    rng = np.random.default_rng(2026)
    md = 20_000
    dfs = [9000, 6000, 4000, 2500, 1200, 150, 90, 40]
    lists, positions = [], []
    for t, df in enumerate(dfs):
        docs = np.sort(rng.choice(md, size=df, replace=False))
        tfs = rng.integers(1, 4, size=df)
        for j, big in enumerate((255, 256, 300, 1000)):
            tfs[(7 * t + 31 * j) % df] = big
        tfs[: min(df, 4)] = [254, 255, 3, 256][: min(df, 4)]  # one group of four holding both sides of 255
        lists.append(list(zip(docs.tolist(), tfs.tolist())))
        if with_positions:
            positions.append([sorted(rng.choice(4000, size=int(tf), replace=False).tolist()) for tf in tfs])
    if with_positions:  # plant the phrase "0 1 2 3" (positions 100..103) in 30 docs that hold all four terms
        common = sorted(set.intersection(*[set(d for d, _ in lists[t]) for t in range(4)]))[:30]
        for t in range(4):
            at = {d: i for i, (d, _) in enumerate(lists[t])}
            for d in common:
                ps = [x for x in positions[t][at[d]] if not 100 <= x <= 103]
                ps = sorted(ps[: len(positions[t][at[d]]) - 1] + [100 + t])
                positions[t][at[d]] = ps
    fieldnorms = rng.integers(1200, 4000, size=md).tolist()
    return O.build_segment(md, lists, fieldnorms,
                           record_option=O.WITH_FREQS_AND_POSITIONS if with_positions else O.WITH_FREQS,
                           positions=positions if with_positions else None)


@pytest.mark.parametrize("k", [3, 10, 100])
def test_union_with_saturated_tf_bytes(ta, k):
    """tf >= 255 does not fit the byte-wide tfs (TqdTerm::tf8): the shared-union kernel has to read the
    packed value for exactly those postings.  Pruned == exhaustive == oracle."""
    seg = _big_tf_segment(False)
    qs = [(O.MODE_OR, [0, 1, 2, 3, 4]), (O.MODE_OR, [7, 0, 1]), (O.MODE_OR, [5, 6, 7, 2, 4]),
          (O.MODE_OR, [6, 3]), (O.MODE_OR, [4]), (O.MODE_OR, [5, 0]), (O.MODE_OR, [1, 2, 3, 5, 6, 7, 0, 4])]
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 16)  # df >= 1250: lists 0..3 dense; 4 (df 1200) and below not
        got = {}
        for ex in (1, 0):
            dev.set_option("exhaustive", ex)
            got[ex] = dev.search(qs, k)
            if ex == 0:  # the kernel this test is about
                assert dev.last_batch_stats()["kernel_mask"] == ta.binding.KERNEL_USHARE, dev.last_batch_stats()
        for a, b in zip(got[0], got[1]):
            assert np.array_equal(a, b)
        sc, _, docs, cnt = got[0]
        for qi, (mode, terms) in enumerate(qs):
            want = O.search(seg, terms, mode, k, pruned=False)
            g = [(float(sc[qi, j]), int(docs[qi, j])) for j in range(int(cnt[qi]))]
            assert [d for _, d in g] == [d for _, d in want], (terms, g, want)
            for (gs, _), (ws, _) in zip(g, want):
                assert rel_close(gs, ws, 1e-5)
    finally:
        dev.close()


def test_phrase_with_saturated_tf_bytes(ta):
    """The phrase kernels take a posting's tf and the index of its first position from the tf bytes
    of its group of four: a saturated byte in the group must send them to the packed values."""
    seg = _big_tf_segment(True)
    qs = [(O.MODE_PHRASE, [0, 1]), (O.MODE_PHRASE, [1, 0, 2]), (O.MODE_PHRASE, [2, 3]), (O.MODE_PHRASE, [0, 1, 2, 3]),
          (O.MODE_PHRASE, [3, 0])]
    for ratio in (16, 2):  # all lists of the queries dense (sweep kernel) / only lists 0 and 1 dense
        dev = ta.DeviceIndex([seg])
        try:
            dev.set_option("dense_ratio", ratio)
            sc, _, docs, cnt = dev.search(qs, 20)
            for qi, (mode, terms) in enumerate(qs):
                want = O.search(seg, terms, mode, 20, pruned=False)
                g = [(float(sc[qi, j]), int(docs[qi, j])) for j in range(int(cnt[qi]))]
                assert [d for _, d in g] == [d for _, d in want], (ratio, terms, g, want)
                assert terms != [0, 1, 2, 3] or len(want) == 20
                for (gs, _), (ws, _) in zip(g, want):
                    assert rel_close(gs, ws, 1e-5)
        finally:
            dev.close()


def test_large_pruned_batch_is_planned_in_slabs(ta):
    """Batches of >= 4096 queries are turned into descriptors by the planner's threads, a slab of
    queries each, and laid end to end (tq_api.cpp: plan_query / QuerySlab); unpruned batches are
    planned by the calling thread.  Every kind of query in one batch of 6000: the two plans must
    give the same answers, bit for bit, and a sample of them the oracle's."""
    seg = O.synth_segment(400_000, n_terms=64, with_positions=True, phrase_terms=16)
    rng = np.random.default_rng(12)
    M, S, N = O.MUST, O.SHOULD, O.MUST_NOT
    a = O.zipf_queries(3000, 2, 64, seed=41)
    o = O.zipf_queries(1500, 4, 64, seed=42)
    queries = []
    for i in range(6000):
        kind = i % 4
        if kind in (0, 2):
            queries.append((O.MODE_AND, a[i // 2].tolist()))
        elif kind == 1:
            queries.append((O.MODE_OR, o[i // 4].tolist()))
        elif i % 8 == 3:
            t = rng.choice(16, size=2, replace=False).tolist()
            queries.append((O.MODE_PHRASE, t, [0, 1]))
        else:
            t = rng.choice(64, size=3, replace=False).tolist()
            queries.append((ta.MODE_BOOL, t, [M, S, N]))
    dev = ta.DeviceIndex([seg])
    try:
        pr = dev.search(queries, 10)
        dev.set_option("exhaustive", 1)
        ex = dev.search(queries, 10)
        for x, y in zip(pr, ex):
            assert np.array_equal(x, y)
        sc, _, docs, cnt = pr
        for qi in range(0, 6000, 187):
            q = queries[qi]
            if q[0] == ta.MODE_BOOL:
                continue
            want = O.search(seg, q[1], q[0], 10, pruned=False, phrase_offsets=q[2] if q[0] == O.MODE_PHRASE else None)
            g = [(float(sc[qi, j]), int(docs[qi, j])) for j in range(int(cnt[qi]))]
            assert [d for _, d in g] == [d for _, d in want], (q, g, want)
            for (gs, _), (ws, _) in zip(g, want):
                assert rel_close(gs, ws, 1e-5)
    finally:
        dev.close()
