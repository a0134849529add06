"""GPU parity of the doc-major union launch (tantivy_amd/csrc/tq_xunion.hip): unions scored WITHOUT
pruning (every match visited — tantivy under a (TopDocs, Count) pair), a whole batch against tiles
of 128 docs.  Doc ids and counts bit-exact against the oracle's exhaustive executor, scores within
1e-5 (3+ term sums; BASELINE.json), and bit-equal to the pruned kernels' scores."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rel_close
from tests.test_gpu_round3 import _alive_bytes, _big_tf_segment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


@pytest.fixture(scope="module")
def seg300k():
    return O.synth_segment(300_000, n_terms=64)


def _or_stream(n, n_terms_per_query, max_rank, seed):
    return [(O.MODE_OR, t.tolist()) for t in O.zipf_queries(n, n_terms_per_query, max_rank, seed=seed)]


def _check_against_oracle(seg, queries, got, k, deleted=None):
    sc, _, docs, cnt = got
    for qi, (mode, terms) in enumerate(queries):
        d, s = O.match_all(seg, terms, mode)
        if deleted is not None and len(d):
            keep = ~np.isin(d, deleted)
            d, s = d[keep], s[keep]
        order = np.lexsort((d, -s.astype(np.float64)))[:k]
        want = [(float(s[i]), int(d[i])) for i in order]
        g = [(float(sc[qi, j]), int(docs[qi, j])) for j in range(int(cnt[qi]))]
        assert [x for _, x in g] == [x for _, x in want], (terms, g[:5], want[:5])
        for (gs, _), (ws, _) in zip(g, want):
            assert rel_close(gs, ws, 1e-5)


@pytest.mark.parametrize("k", [10, 100])
def test_unpruned_unions_run_doc_major_and_match_the_oracle(ta, seg300k, k):
    """96 five-term Zipf unions + the edge shapes (one list, eight lists, a list twice, rare lists
    only): lists with a bitmap and lists kept as plain arrays meet in the same tiles.  The launch's
    tile count says which kernel ran."""
    seg = seg300k
    queries = _or_stream(96, 5, 64, 31) + [(O.MODE_OR, [0]), (O.MODE_OR, [63, 0]), (O.MODE_OR, [1, 1, 2]),
                                           (O.MODE_OR, [0, 1, 2, 3, 4, 5, 6, 7]), (O.MODE_OR, [5, 60, 61, 62, 63])]
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("timing", 1)
        dev.set_option("exhaustive", 1)
        dev.set_option("dense_ratio", 16)  # lists 0..7 get bitmaps, the others become plain arrays
        dev.set_option("xunion_ratio", 1 << 30)  # (whatever the lists hold)
        got = dev.search(queries, k)
        st = dev.last_batch_stats()
        _check_against_oracle(seg, queries, got, k)
        assert st["tiles"] == (seg.max_doc + 127) // 128, st  # every query of the batch was eligible
        assert st["kernel_mask"] == ta.binding.KERNEL_XUNION, st
        # every match was visited: the per-query counts are the union sizes
        counts = dev.last_batch_match_counts(len(queries))
        for qi, (mode, terms) in enumerate(queries):
            assert int(counts[qi]) == len(O.match_all(seg, terms, mode)[0]), terms
        # the pruned kernels return the same bits
        dev.set_option("exhaustive", 0)
        pr = dev.search(queries, k)
        assert dev.last_batch_stats()["kernel_mask"] & ta.binding.KERNEL_USHARE  # (the shared-union launch)
        for a, b in zip(got, pr):
            assert np.array_equal(a, b)
    finally:
        dev.close()


def test_unpruned_unions_with_deletes_and_count(ta, seg300k):
    seg = seg300k
    rng = np.random.default_rng(77)
    dele = np.sort(rng.choice(seg.max_doc, size=seg.max_doc // 4, replace=False))
    queries = _or_stream(64, 3, 64, 5) + _or_stream(32, 2, 16, 6)
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_alive_bitset(_alive_bytes(seg.max_doc, dele.tolist()))
        dev.set_option("exhaustive", 1)
        got = dev.search(queries, 10)
        _check_against_oracle(seg, queries, got, 10, deleted=dele)
        cnt = dev.count(queries)
        for qi, (mode, terms) in enumerate(queries):
            d = O.match_all(seg, terms, mode)[0]
            assert int(cnt[qi]) == int((~np.isin(d, dele)).sum()), terms
    finally:
        dev.close()


@pytest.mark.parametrize("k", [3, 100])
def test_unpruned_unions_with_saturated_tf_bytes(ta, k):
    """tf >= 255 in lists with a bitmap and in lists kept as plain arrays: the byte says "read the
    packed value"."""
    seg = _big_tf_segment(False)
    base = [(O.MODE_OR, [0, 1, 2, 3, 4]), (O.MODE_OR, [7, 0, 1]), (O.MODE_OR, [5, 6, 7, 2, 4]),
            (O.MODE_OR, [5, 0]), (O.MODE_OR, [1, 2, 3, 5, 6, 7, 0, 4]), (O.MODE_OR, [6, 3])]
    qs = base * 4
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 16)  # lists 0..3 get bitmaps, 4..7 do not
        dev.set_option("xunion_min_queries", 16)
        dev.set_option("xunion_ratio", 6)  # ([6, 3] keeps the window kernel)
        dev.set_option("exhaustive", 1)
        got = dev.search(qs, k)
        _check_against_oracle(seg, qs, got, k)
        dev.set_option("exhaustive", 0)
        pr = dev.search(qs, k)
        for a, b in zip(got, pr):
            assert np.array_equal(a, b)
    finally:
        dev.close()


def test_small_and_mixed_batches_keep_the_window_kernel(ta, seg300k):
    """Below 64 eligible queries, or with lists too sparse to pay for a pass over every doc, the
    batch stays with the per-query window kernel; AND queries of the same batch are untouched."""
    seg = seg300k
    queries = _or_stream(6, 5, 64, 9) + [(O.MODE_AND, [0, 1]), (O.MODE_OR, [62, 63])]
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("exhaustive", 1)
        got = dev.search(queries, 10)
        st = dev.last_batch_stats()
        assert not (st["kernel_mask"] & ta.binding.KERNEL_XUNION) and st["kernel_mask"] & ta.binding.KERNEL_OR_WINDOWS, st
        _check_against_oracle(seg, queries, got, 10)
    finally:
        dev.close()
