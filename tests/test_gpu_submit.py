"""GPU tests of the concurrent single-query entry (tq_submit / tq_wait / tq_search_one,
include/tantivy_amd.h): tantivy lets any number of threads call Searcher::search at once, one query
per call (src/core/searcher.rs:180-238, src/collector/mod.rs:173-183).  Here T host threads issue
single queries against one segment; the library coalesces them into batched launches.  Every result
must equal the oracle's (bit-exact docs, 2-term scores bit-equal) and the batched path's."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rel_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


@pytest.fixture(scope="module")
def seg300k():
    return O.synth_segment(300_000, n_terms=64)


def _mixed_queries(n, seed):
    a = O.zipf_queries(n // 2, 2, 64, seed=seed)
    o = O.zipf_queries(n - n // 2, 3, 64, seed=seed + 1)
    return [(O.MODE_AND, a[i // 2].tolist()) if i % 2 == 0 else (O.MODE_OR, o[i // 2].tolist()) for i in range(n)]


def _check(seg, queries, got, k):
    sc, _, docs, cnt = got[:4]
    for qi, (mode, terms) in enumerate(queries):
        d, s = O.match_all(seg, terms, mode)
        order = np.lexsort((d, -s.astype(np.float64)))[:k]
        want = [(float(s[i]), int(d[i])) for i in order]
        g = [(float(sc[qi, j]), int(docs[qi, j])) for j in range(int(cnt[qi]))]
        assert [x for _, x in g] == [x for _, x in want], (terms, g[:4], want[:4])
        for (gs, _), (ws, _) in zip(g, want):
            if mode == O.MODE_AND and len(terms) == 2:
                assert np.float32(gs) == np.float32(ws)
            else:
                assert rel_close(gs, ws, 1e-5)


@pytest.mark.parametrize("n_threads", [1, 32])
def test_concurrent_single_queries_equal_the_oracle(ta, seg300k, n_threads):
    """32 threads x 200 single queries (AND / OR mixed) through Searcher::search on one segment."""
    seg = seg300k
    queries = _mixed_queries(200 * 32 if n_threads > 1 else 96, 404)
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 64)
        dev.submit_stats(reset=True)
        got = dev.search_concurrent(queries, 10, n_threads)
        st = dev.submit_stats()
        _check(seg, queries, got, 10)
        assert st["queries"] == len(queries), st
        if n_threads > 1:  # the calls were coalesced: far fewer launches than queries
            assert st["batches"] < len(queries) // 4 and st["max_batch"] > 8, st
        else:
            assert st["batches"] == len(queries) and st["max_batch"] == 1, st
        batched = dev.search(queries, 10)
        for a, b in zip(got[:4], batched):
            assert np.array_equal(a, b)
    finally:
        dev.close()


def test_concurrent_queries_over_two_segments_and_an_unsupported_one(ta):
    """Two segments (global statistics, merge_top_k per call); a query the device does not take
    fails alone — its batch mates still get their results."""
    segs = [O.synth_segment(120_000, n_terms=32, segment_ord=o) for o in range(2)]
    queries = [(O.MODE_AND, t.tolist()) for t in O.zipf_queries(600, 2, 32, seed=9)]
    dev = ta.DeviceIndex(segs, devices=[0])
    try:
        got = dev.search_concurrent(queries, 5, 16)
        want = dev.search(queries, 5)
        for a, b in zip(got[:4], want):
            assert np.array_equal(a, b)
        # raw ABI: 24 threads, one of them submits a phrase on a segment without positions
        L = ta.binding.lib()
        raw = dev.segment_raw(0)
        cache = np.ascontiguousarray(ta.bm25_for_terms([1000], 240_000, 240_000 * 20)[1], np.float32)
        results, errors = {}, {}

        def worker(i):
            import ctypes as C
            terms = [i % 8, 8 + i % 16]
            hs = (C.c_uint32 * 2)(*[dev.term_handle(t, 0) for t in terms])
            ws = (C.c_float * 2)(2.0, 1.0)
            q = ta.binding.TqQuery()
            q.n_terms = 2
            q.terms = C.cast(hs, C.POINTER(C.c_uint32))
            q.weights = C.cast(ws, C.POINTER(C.c_float))
            q.tf_cache = cache.ctypes.data_as(C.POINTER(C.c_float))
            q.k = 3
            q.mode = O.MODE_AND
            offs = (C.c_uint32 * 2)(0, 1)
            if i == 7:  # a phrase: this segment has no positions
                q.mode = O.MODE_PHRASE
                q.phrase_offsets = C.cast(offs, C.POINTER(C.c_uint32))
            sc = np.zeros(3, np.float32)
            dc = np.zeros(3, np.uint32)
            ct = np.zeros(1, np.uint32)
            rc = L.tq_search_one(raw, C.byref(q), None, sc.ctypes.data_as(C.POINTER(C.c_float)),
                                 dc.ctypes.data_as(C.POINTER(C.c_uint32)), ct.ctypes.data_as(C.POINTER(C.c_uint32)))
            if rc:
                errors[i] = (rc, L.tq_last_error().decode())
            else:
                results[i] = (sc.copy(), dc.copy(), int(ct[0]))

        th = [threading.Thread(target=worker, args=(i,)) for i in range(24)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert list(errors) == [7] and errors[7][0] == 4, errors  # TQ_ERR_UNSUPPORTED, alone
        assert len(results) == 23
        for i, (sc, dc, ct) in results.items():
            terms = [i % 8, 8 + i % 16]
            d, _ = O.match_all(segs[0], terms, O.MODE_AND)
            assert ct == min(3, len(d)) and set(dc[:ct].tolist()) <= set(d.tolist())
    finally:
        dev.close()


def test_hundreds_of_callers_per_batch_round_after_round(ta, seg300k):
    """The waiting side of tq_search_one as rebuilt in round 6 (a ticket's caller sleeps under the ticket's own mutex,
    is unlinked and woken by the batch's leader off the queue's lock, submit + park take the queue's mutex once):
    512 threads x 24 queries, three rounds on one segment — every row equal to the batch path's, every query answered
    exactly once, batches of hundreds of callers."""
    seg = seg300k
    queries = _mixed_queries(512 * 24, 405)
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 64)
        batched = dev.search(queries, 10)
        for _ in range(3):
            dev.submit_stats(reset=True)
            got = dev.search_concurrent(queries, 10, 512)
            st = dev.submit_stats()
            assert st["queries"] == len(queries), st
            assert st["max_batch"] >= 64, st
            for a, b in zip(got[:4], batched):
                assert np.array_equal(a, b)
        _check(seg, queries[:200], got, 10)
    finally:
        dev.close()
