"""GPU parity of boolean queries in the shared leader-major launch (tq_ashare.hip, boolean leads): one
lead per (query, list of its lead set); a leader block is decoded once for every boolean query it
leads in the batch, the doc-matrix word of a doc says which of the query's other lists can hold it
(its own score bound, the lists the scoring stage probes).  BooleanWeight::complex_scorer semantics
(src/query/boolean_query/boolean_weight.rs:236-431: RequiredOptionalScorer, Exclude, Intersection of
unions): docs bit-exact against the oracle, and the pruned shared launch returns the bits of the
exhaustive per-query union kernel (tq_union.hip).  Every test asserts WHICH kernel ran."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_gpu_parity import _assert_bool_hits, _bool_want, _device_topk
from tests.test_gpu_round3 import _alive_bytes

pytestmark = pytest.mark.gpu

M, S, N = O.MUST, O.SHOULD, O.MUST_NOT
# the shapes of test_boolean_mixed_occurs (tests/test_gpu_parity.py) + the bench's union_intersection
# group (benches/and_or_queries.rs:150-153): (occurs, clause_of, minimum_number_should_match)
SHAPES = [([M, S], None, 0), ([M, N], None, 0), ([S, N], None, 0), ([M, M, S], None, 0), ([M, S, S], None, 0),
          ([S, S, N], None, 0), ([M, M, M, N], None, 0), ([M, S, N, S, M], None, 0), ([M, M, M, M, S], None, 0),
          ([N, S], None, 0), ([N, N, M], None, 0), ([M, S, N], None, 0),
          ([M, M, M], [0, 1, 1], 0), ([M, M, M, M], [0, 0, 1, 1], 0), ([M, M, M], [0, 0, 1], 0),
          ([M, M, M, S], [0, 1, 1, 2], 0), ([M, M, M, N], [0, 0, 1, 2], 0), ([M, M, M, M, M], [0, 1, 1, 2, 2], 0),
          ([M, M, M, M], [0, 1, 1, 1], 0), ([M, S, S], None, 1), ([M, S, S, S], None, 2), ([S, S, S], None, 2),
          ([S, S, S], None, 3), ([S, S, N], None, 2), ([M, S], None, 1), ([S, S, S, S, N], None, 3),
          ([M, S, S, N], [0, 1, 1, 2], 1), ([N, N, S, S], [0, 0, 1, 2], 0),
          ([M, M, S, S, S, N, N, M], None, 0)]


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def _all_dense(dev):
    dev.set_option("dense_ratio", 4096)    # every list of the test vocabularies gets a bitmap ...
    dev.set_option("dense_budget_x", 256)  # ... whatever the segment's size


def _queries(ta, rng, n_terms, reps):
    out = []
    for occ, cof, msm in SHAPES * reps:
        terms = rng.choice(n_terms, size=len(occ), replace=False).tolist()
        out.append((ta.MODE_BOOL, terms, list(occ), cof, msm))
    return out


@pytest.mark.parametrize("seed", [21, 22])
def test_boolean_shapes_in_the_shared_launch(ta, seed):
    """Every shape x random lists of a 48-list vocabulary (all with bitmaps; 40 get doc-matrix columns, the
    others signature bits): pruned through the shared launch == oracle == the exhaustive per-query union
    kernel's docs, with and without deletes, for k below and above 16 and 64 (slot rows, registers)."""
    rng = np.random.default_rng(seed)
    seg = O.synth_segment(150_000 + 10_000 * seed, n_terms=48, with_positions=False)
    queries = _queries(ta, rng, 48, 4)
    deleted = set(rng.choice(seg.max_doc, size=seg.max_doc // 6, replace=False).tolist())
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("timing", 1)
        _all_dense(dev)
        for dels in ((), deleted):
            dev.set_alive_bitset(_alive_bytes(seg.max_doc, dels) if dels else None)
            want = [_bool_want(seg, q[1], q[2], dels, q[3], q[4]) for q in queries]
            for k in (1, 10, 17, 100):
                dev.set_option("exhaustive", 0)
                got = _device_topk(dev, queries, k)
                st = dev.last_batch_stats()
                assert st["kernel_mask"] & ta.binding.KERNEL_BSHARE, st
                dev.set_option("exhaustive", 1)
                ex = _device_topk(dev, queries, k)
                st_e = dev.last_batch_stats()
                assert not (st_e["kernel_mask"] & ta.binding.KERNEL_BSHARE), st_e
                for q, g, e, w in zip(queries, got, ex, want):
                    try:
                        _assert_bool_hits(g, w, k, q[2], q[3])
                        assert [d for _, d in g] == [d for _, d in e] or len(w) > k  # (near-ties across rank k)
                    except AssertionError:
                        raise AssertionError("seed %d k %d deletes %d query %r\ngot  %r\nwant %r\nexh  %r" %
                                             (seed, k, bool(dels), q, g[:5], w[:5], e[:5]))
    finally:
        dev.close()


def test_zipf_stream_of_the_bench_shapes_shares_leaders(ta):
    """1 200 boolean queries of the bench's four shapes over the 24 most frequent of 64 lists: a few
    hundred (query, leading list) pairs per leader, groups of 32, twins (repeated queries) — bit-equal
    to the run with the shared launch switched off, docs equal to the exhaustive run's."""
    seg = O.synth_segment(300_000, n_terms=64, with_positions=False)
    ids = O.zipf_queries(1200, 4, 24, seed=20260924)
    shapes = [(3, [M, M, M], [0, 1, 1]), (4, [M, M, M, M], [0, 0, 1, 1]), (3, [M, S, N], None), (3, [M, M, M], [0, 0, 1])]
    queries = []
    for i, q in enumerate(ids):
        nt, occ, cof = shapes[i % 4]
        queries.append((ta.MODE_BOOL, q.tolist()[:nt], occ, cof, 0))
    queries += queries[:100]  # twins
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("timing", 1)
        _all_dense(dev)
        dev.set_option("exhaustive", 0)
        pr = dev.search(queries, 10)
        st = dev.last_batch_stats()
        assert st["kernels"] == ["bshare"], st
        dev.set_option("exhaustive", 1)
        ex = dev.search(queries, 10)
        assert np.array_equal(pr[3], ex[3])
        for qi in range(len(queries)):
            c = int(pr[3][qi])
            assert np.array_equal(pr[2][qi, :c], ex[2][qi, :c]), (queries[qi], pr[2][qi], ex[2][qi])
            assert np.array_equal(pr[0][qi, :c], ex[0][qi, :c]), (queries[qi], pr[0][qi], ex[0][qi])
        for qi in range(0, len(queries), 37):  # a sample against the oracle
            q = queries[qi]
            got = [(float(pr[0][qi, j]), int(pr[2][qi, j])) for j in range(int(pr[3][qi]))]
            _assert_bool_hits(got, _bool_want(seg, q[1], q[2], (), q[3], q[4]), 10, q[2], q[3])
    finally:
        dev.close()


def test_lists_below_dense_ratio_get_probe_tables_on_first_use(ta):
    """A boolean query naming a list without a bitmap: with "probe_budget_x" 0 and "rdir_budget_x" 0 it stays on the
    union kernel (the same batch's other boolean queries take the shared launch); with the default probe budget the
    list gets a bitmap + tf bytes of its own the first time a boolean query names it, with no probe budget but range
    directories (round 6) the shared launch probes those, and every query is shared — the other kernels keep seeing the
    list as sparse (no new dense list in the segment's stats).  Results as the oracle's, and identical between the
    settings."""
    rng = np.random.default_rng(5)
    seg = O.synth_segment(200_000, n_terms=64, with_positions=False)
    queries = _queries(ta, rng, 64, 2)
    easy = [(ta.MODE_BOOL, [1, 2, 3], [M, S, N], None, 0), (ta.MODE_BOOL, [4, 2, 0], [M, M, M], [0, 1, 1], 0)] * 4
    want = [_bool_want(seg, q[1], q[2], (), q[3], q[4]) for q in queries + easy]
    got = {}
    for budget, rdir in ((0, 0), (16, 0), (0, 4)):
        dev = ta.DeviceIndex([seg])
        try:
            dev.set_option("timing", 1)
            dev.set_option("dense_ratio", 32)  # lists 0..14 get a bitmap
            dev.set_option("probe_budget_x", budget)
            dev.set_option("rdir_budget_x", rdir)
            dev.set_option("exhaustive", 0)
            got[(budget, rdir)] = _device_topk(dev, queries + easy, 10)
            st = dev.last_batch_stats()
            assert st["kernel_mask"] & ta.binding.KERNEL_BSHARE, st
            if budget == 0 and rdir == 0:
                assert st["kernel_mask"] & ta.binding.KERNEL_BOOL, st
            else:
                assert not (st["kernel_mask"] & ta.binding.KERNEL_BOOL), st
            assert dev.segment_stats(0)["n_dense_lists"] <= 16
            for q, g, w in zip(queries + easy, got[(budget, rdir)], want):
                _assert_bool_hits(g, w, 10, q[2], q[3])
            # an intersection over the same sparse lists in a small batch still runs on the general AND kernel
            dev.search([(O.MODE_AND, [40, 50])] * 8, 10)
            assert dev.last_batch_stats()["kernels"] == ["and"], dev.last_batch_stats()
        finally:
            dev.close()
    assert [[d for _, d in g] for g in got[(0, 0)]] == [[d for _, d in g] for g in got[(0, 4)]]
    got = {0: got[(0, 0)], 16: got[(16, 0)]}
    assert [[d for _, d in g] for g in got[0]] == [[d for _, d in g] for g in got[16]]
