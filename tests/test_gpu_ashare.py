"""GPU parity of the shared-intersection launch (tantivy_amd/csrc/tq_ashare.hip): the AND queries of a
batch driven leader by leader — a leader block decoded once for all the queries that lead with it.
block_wand_intersection semantics (src/query/boolean_query/block_wand_intersection.rs:19-179): doc
ids bit-exact against the oracle, scores bit-equal for 2-term queries (1e-5 for 3+ terms), and the
pruned shared launch returns the bits of the exhaustive per-query kernel (tq_and.hip).  Every test
asserts WHICH kernel ran (tq_batch_stats.kernel_mask)."""
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


def _and_stream(n, n_terms_per_query, max_rank, seed):
    return [(O.MODE_AND, t.tolist()) for t in O.zipf_queries(n, n_terms_per_query, max_rank, seed=seed)]


def _check_against_oracle(seg, queries, got, k, deleted=None, exact2=True):
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
            if exact2 and mode == O.MODE_AND and len(terms) == 2:
                assert np.float32(gs) == np.float32(ws), (terms, g[:5], want[:5])
            else:
                assert rel_close(gs, ws, 1e-5)


def _both_modes(ta, dev, queries, k):
    """(pruned result, its stats, exhaustive result, its stats) of one batch."""
    dev.set_option("exhaustive", 0)
    pr = dev.search(queries, k)
    st_p = dev.last_batch_stats()
    dev.set_option("exhaustive", 1)
    ex = dev.search(queries, k)
    st_e = dev.last_batch_stats()
    return pr, st_p, ex, st_e


@pytest.mark.parametrize("k", [1, 10, 17, 100])
def test_shared_intersections_match_the_oracle_and_the_per_query_kernel(ta, seg300k, k):
    """400 Zipf-sampled 2-term ANDs over 64 lists (32 with bitmaps at dense_ratio 64): the queries
    whose leader leads >= 4 of them and whose other list has a bitmap take the shared launch, the
    others the per-query kernels — same batch, same bits as the exhaustive per-query run."""
    seg = seg300k
    queries = _and_stream(400, 2, 64, 11) + [(O.MODE_AND, [3, 3]), (O.MODE_AND, [0, 63]), (O.MODE_AND, [63, 0])] * 2
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("timing", 1)
        dev.set_option("dense_ratio", 64)
        pr, st_p, ex, st_e = _both_modes(ta, dev, queries, k)
        assert st_p["kernel_mask"] & ta.binding.KERNEL_ASHARE, st_p
        assert not (st_e["kernel_mask"] & ta.binding.KERNEL_ASHARE), st_e
        for a, b in zip(pr, ex):
            assert np.array_equal(a, b)
        _check_against_oracle(seg, queries, pr, k)
        assert st_p["unique_bytes"] <= st_p["algorithmic_bytes"]
    finally:
        dev.close()


def test_every_query_of_a_dense_batch_takes_the_shared_launch(ta, seg300k):
    """Ranks 0..15 only (all lists with bitmaps and doc-matrix columns): nothing is left for the
    per-query kernels, and a second identical batch (thresholds start from zero again) gives the
    same bits."""
    seg = seg300k
    queries = _and_stream(600, 2, 16, 3)
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("dense_ratio", 64)
        dev.set_option("exhaustive", 0)
        a = dev.search(queries, 10)
        st = dev.last_batch_stats()
        assert st["kernel_mask"] == ta.binding.KERNEL_ASHARE, st
        b = dev.search(queries, 10)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        _check_against_oracle(seg, queries, a, 10)
    finally:
        dev.close()


@pytest.mark.parametrize("n_terms", [3, 4, 8])
def test_shared_intersections_of_three_and_more_lists(ta, seg300k, n_terms):
    seg = seg300k
    queries = _and_stream(300, n_terms, 24, 40 + n_terms)
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("dense_ratio", 64)
        pr, st_p, ex, _ = _both_modes(ta, dev, queries, 10)
        assert st_p["kernel_mask"] & ta.binding.KERNEL_ASHARE, st_p
        for a, b in zip(pr, ex):
            assert np.array_equal(a, b)
        _check_against_oracle(seg, queries, pr, 10)
    finally:
        dev.close()


def test_lists_without_a_column_use_their_signature_bit(ta, seg300k):
    """Every list gets a bitmap (dense_ratio 4096) but only the 40 densest a doc-matrix column: pairs
    among ranks 40..63 meet in stage F through their signature bits ('maybe'), the bitmap word
    decides in stage C.  Next to them pairs with a column list, so both kinds share leader groups."""
    seg = seg300k
    queries = [(O.MODE_AND, [i, j]) for i in range(40, 52) for j in range(52, 64)]
    queries += [(O.MODE_AND, [i, j]) for i in range(0, 40, 3) for j in range(52, 64, 2)]
    queries += [(O.MODE_AND, [45, 50, 60]), (O.MODE_AND, [41, 2, 58]), (O.MODE_AND, [40, 41, 42, 63])] * 4
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("dense_ratio", 4096)
        pr, st_p, ex, _ = _both_modes(ta, dev, queries, 10)
        assert st_p["kernel_mask"] == ta.binding.KERNEL_ASHARE, st_p
        assert dev.segment_stats()["n_docmat_columns"] <= 40  # (columns are reserved for the 40 densest lists)
        for a, b in zip(pr, ex):
            assert np.array_equal(a, b)
        _check_against_oracle(seg, queries, pr, 10)
    finally:
        dev.close()


def test_shared_intersections_with_deletes(ta, seg300k):
    seg = seg300k
    rng = np.random.default_rng(99)
    dele = np.sort(rng.choice(seg.max_doc, size=seg.max_doc // 3, replace=False))
    queries = _and_stream(300, 2, 32, 21) + _and_stream(100, 3, 16, 22)
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("dense_ratio", 64)
        dev.set_alive_bitset(_alive_bytes(seg.max_doc, dele.tolist()))
        pr, st_p, ex, _ = _both_modes(ta, dev, queries, 10)
        assert st_p["kernel_mask"] & ta.binding.KERNEL_ASHARE, st_p
        for a, b in zip(pr, ex):
            assert np.array_equal(a, b)
        _check_against_oracle(seg, queries, pr, 10, deleted=dele)
    finally:
        dev.close()


@pytest.mark.parametrize("k", [3, 100])
def test_shared_intersections_with_saturated_tf_bytes(ta, k):
    """tf >= 255: the byte-wide tf of the other list says 'read the packed value'."""
    seg = _big_tf_segment(False)
    base = [(O.MODE_AND, [0, 1]), (O.MODE_AND, [1, 2]), (O.MODE_AND, [2, 0]), (O.MODE_AND, [3, 1]),
            (O.MODE_AND, [4, 0]), (O.MODE_AND, [4, 1, 2]), (O.MODE_AND, [3, 2, 0, 1]), (O.MODE_AND, [4, 3])]
    qs = base * 6
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("dense_ratio", 32)  # lists 0..4 get bitmaps
        pr, st_p, ex, _ = _both_modes(ta, dev, qs, k)
        assert st_p["kernel_mask"] & ta.binding.KERNEL_ASHARE, st_p
        for a, b in zip(pr, ex):
            assert np.array_equal(a, b)
        _check_against_oracle(seg, qs, pr, k)
    finally:
        dev.close()


def test_shared_intersections_next_to_shared_unions(ta, seg300k):
    """The mixed stream of config 5: both term-major launches in one batch (two persistent grids on
    two streams), every result equal to the exhaustive per-query kernels' and the oracle's."""
    seg = seg300k
    a = _and_stream(300, 2, 32, 5)
    o = [(O.MODE_OR, t.tolist()) for t in O.zipf_queries(300, 5, 64, seed=6)]
    queries = [x for pair in zip(a, o) for x in pair]
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        dev.set_option("dense_ratio", 64)
        pr, st_p, ex, _ = _both_modes(ta, dev, queries, 10)
        assert st_p["kernel_mask"] & ta.binding.KERNEL_ASHARE and st_p["kernel_mask"] & ta.binding.KERNEL_USHARE, st_p
        for x, y in zip(pr, ex):
            assert np.array_equal(x, y)
        _check_against_oracle(seg, queries, pr, 10)
    finally:
        dev.close()


def test_full_size_shared_intersections(ta):
    """10M docs, the bench's vocabulary: 2000 queries of the headline stream, pruned shared launch ==
    exhaustive per-query kernel, 24 of them against the oracle."""
    seg = O.synth_segment(10_000_000, n_terms=256)
    queries = _and_stream(2000, 2, 256, 20260921)
    dev = ta.DeviceIndex([seg])
    dev.set_option("ashare_min_batch", 0)  # (whatever the batch size)
    try:
        pr, st_p, ex, _ = _both_modes(ta, dev, queries, 10)
        assert st_p["kernel_mask"] & ta.binding.KERNEL_ASHARE, st_p
        for a, b in zip(pr, ex):
            assert np.array_equal(a, b)
        sample = list(range(0, 2000, 83))
        sub = [queries[i] for i in sample]
        got = tuple(x[sample] for x in pr)
        _check_against_oracle(seg, sub, got, 10)
    finally:
        dev.close()


def test_small_batches_keep_the_per_query_kernels(ta, seg300k):
    """"ashare_min_batch" (default 16 qualifying queries; 512 until round 6): below it the batch's intersections stay
    on the per-query kernels (two launches and per-task set-up cost more than sharing saves); same results either
    way."""
    queries = _and_stream(400, 2, 64, 13)
    dev = ta.DeviceIndex([seg300k])
    try:
        dev.set_option("timing", 1)
        dev.set_option("dense_ratio", 64)
        dev.set_option("exhaustive", 0)
        small = dev.search(queries[:12], 10)
        st = dev.last_batch_stats()
        assert not (st["kernel_mask"] & ta.binding.KERNEL_ASHARE), st
        dev.set_option("ashare_min_batch", 500)
        a = dev.search(queries, 10)
        st = dev.last_batch_stats()
        assert not (st["kernel_mask"] & ta.binding.KERNEL_ASHARE), st
        dev.set_option("ashare_min_batch", 300)
        b = dev.search(queries, 10)
        st = dev.last_batch_stats()
        assert st["kernel_mask"] & ta.binding.KERNEL_ASHARE, st
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        for x, y in zip(small, b):
            assert np.array_equal(x, y[:12])
    finally:
        dev.close()


def test_two_list_intersections_probe_a_list_without_a_bitmap(ta, seg300k):
    """dense_ratio 8: three lists have bitmaps.  A batch of >= ashare_min_batch 2-term intersections runs entirely in
    the shared launch: the lists it probes (the more frequent list of each query) have range directories from their
    preparation ("rdir_budget_x", round 6) — or, without those, get bitmap + tf bytes built on first use
    ("probe_budget_x"); with both budgets at 0 the same batch keeps the general per-query kernel for those queries.
    Same bits every way, equal to the oracle's."""
    queries = _and_stream(1300, 2, 64, 17)
    got = {}
    for probe, rdir in ((0, 0), (16, 0), (0, 4)):
        dev = ta.DeviceIndex([seg300k])
        try:
            dev.set_option("timing", 1)
            dev.set_option("dense_ratio", 8)
            dev.set_option("probe_budget_x", probe)
            dev.set_option("rdir_budget_x", rdir)
            dev.set_option("ashare_min_batch", 200)
            dev.set_option("exhaustive", 0)
            got[(probe, rdir)] = dev.search(queries, 10)
            st = dev.last_batch_stats()
            assert st["kernel_mask"] & ta.binding.KERNEL_ASHARE, st
            if probe or rdir:
                assert not (st["kernel_mask"] & ta.binding.KERNEL_AND), st
            else:
                assert st["kernel_mask"] & ta.binding.KERNEL_AND, st
            assert dev.segment_stats(0)["n_dense_lists"] <= 4
        finally:
            dev.close()
    for key in ((16, 0), (0, 4)):
        for a, b in zip(got[(0, 0)], got[key]):
            assert np.array_equal(a, b), key
    got = {16: got[(16, 0)]}
    _check_against_oracle(seg300k, queries[::13], [x[::13] for x in got[16]], 10)
