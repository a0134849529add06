"""GPU parity of nested boolean queries (tantivy_amd/csrc/tq_tree.hip): the
`SpecializedScorer::Other` trees of BooleanWeight::complex_scorer (boolean_weight.rs:236-431) that do not flatten —
an intersection inside a union, under MustNot, nested MustNot / optional terms, minimum_number_should_match inside a
nested query and over nested groups (disjunction.rs:113-139), an intersection one level further down
(`+a +((+b +c) d)`).  Oracle: O.tree_match_all, complex_scorer restated on every level in numpy (its union-only
shapes agree bit for bit with the C scorer tree: tests/test_tree_oracle_cpu.py).  Doc ids exact; scores within 1e-5
(sums of 3+ terms: the reference's own order is not canonical); pruned == exhaustive bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import random_postings, rel_close
from tests.tree_shapes import SHAPES, to_device, to_oracle

pytestmark = pytest.mark.gpu

M, S, N = O.MUST, O.SHOULD, O.MUST_NOT


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


@pytest.fixture(scope="module")
def seg():
    return O.synth_segment(300_000, n_terms=48)


def _check(ta, dev, seg, specs, k, deleted=None):
    queries = [to_device(ta, sp, msm) for sp, msm in specs]
    dev.set_option("timing", 1)
    out = {}
    for mode in (0, 1):
        dev.set_option("exhaustive", mode)
        out[mode] = dev.search(queries, k)
        st = dev.last_batch_stats()
        assert st["kernel_mask"] & ta.binding.KERNEL_TREE, st
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    sc, _, dc, ct = out[0]
    for i, (sp, msm) in enumerate(specs):
        want = O.tree_search(seg, to_oracle(sp), k, msm, deleted)
        got = [(float(sc[i, j]), int(dc[i, j])) for j in range(int(ct[i]))]
        assert len(got) == len(want), (sp, msm, got, want)
        if [d for _, d in got] != [d for _, d in want]:  # near-ties of 3+ term sums may swap neighbours
            assert sorted(d for _, d in got) == sorted(d for _, d in want), (sp, msm, got, want)
        for (gs, _), (ws, _) in zip(sorted(got, key=lambda x: x[1]), sorted(want, key=lambda x: x[1])):
            assert rel_close(gs, ws, 1e-5), (sp, msm, got, want)
    return out[0]


def test_nested_shapes_against_the_tree_oracle(ta, seg):
    """every shape of tests/tree_shapes.py over dense and sparse lists (probe tables are built on first use)"""
    rng = np.random.default_rng(3)
    specs = []
    for shape, msm in SHAPES:
        for _ in range(3):
            ids = rng.permutation(40)[:8].tolist()
            specs.append((shape(ids), msm))
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 32)
        for k in (1, 10, 100):
            _check(ta, dev, seg, specs, k)
        # the lists below dense_ratio got probe tables; the per-query kernels still see them as sparse
        assert dev.segment_stats(0)["n_dense_lists"] <= 16
    finally:
        dev.close()


def test_nested_queries_with_deletes_and_a_mixed_batch(ta, seg):
    """nested queries next to intersections, unions and flat boolean queries in ONE batch; a quarter of the docs
    deleted (AliveBitSet: the bitmap expression ANDs the alive words)."""
    rng = np.random.default_rng(4)
    deleted = np.sort(rng.choice(seg.max_doc, size=seg.max_doc // 4, replace=False)).astype(np.uint32)
    alive = np.ones(seg.max_doc, bool)
    alive[deleted] = False
    bits = np.packbits(alive, bitorder="little")
    body = np.uint32(seg.max_doc).tobytes() + bits.tobytes() + b"\0" * ((-len(bits)) % 8)
    specs = [(shape(rng.permutation(24)[:8].tolist()), msm) for shape, msm in SHAPES]
    flat = [(O.MODE_AND, [3, 9]), (O.MODE_OR, [1, 2, 5, 9, 30]), (ta.MODE_BOOL, [1, 2, 3], [M, S, N], None, 0)]
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_alive_bitset(body)
        _check(ta, dev, seg, specs, 10, deleted)
        queries = [to_device(ta, sp, msm) for sp, msm in specs] + flat
        dev.set_option("exhaustive", 0)
        got = dev.search(queries, 10)
        st = dev.last_batch_stats()
        assert st["kernel_mask"] & ta.binding.KERNEL_TREE and bin(st["kernel_mask"]).count("1") >= 3, st
        for i, q in enumerate(flat):
            qi = len(specs) + i
            if q[0] == ta.MODE_BOOL:
                d, s = O.bool_match_all(seg, q[1], q[2])
            else:
                d, s = O.match_all(seg, q[1], q[0])
            keep = ~np.isin(d, deleted)
            d, s = d[keep], s[keep]
            order = np.lexsort((d, -s.astype(np.float64)))[:10]
            assert [int(got[2][qi, j]) for j in range(int(got[3][qi]))] == [int(d[o]) for o in order], q
    finally:
        dev.close()


def test_nested_queries_through_searcher_search_and_count(ta, seg):
    """Searcher::search, one query per call from 8 threads (tq_search_one coalesces them), and the Count collector
    (tq_count_batch: an exhaustive scan of the same kernel) on nested queries."""
    rng = np.random.default_rng(5)
    specs = [(shape(rng.permutation(30)[:8].tolist()), msm) for shape, msm in SHAPES] * 3
    queries = [to_device(ta, sp, msm) for sp, msm in specs]
    dev = ta.DeviceIndex([seg])
    try:
        sc, _, dc, ct, _, _ = dev.search_concurrent(queries, 10, 8)
        for i, (sp, msm) in enumerate(specs):
            want = O.tree_search(seg, to_oracle(sp), 10, msm)
            assert sorted(int(dc[i, j]) for j in range(int(ct[i]))) == sorted(d for _, d in want), (sp, msm)
        counts = dev.count(queries)
        for i, (sp, msm) in enumerate(specs):
            assert int(counts[i]) == len(O.tree_match_all(seg, to_oracle(sp), msm)[0]), (sp, msm)
    finally:
        dev.close()


def test_unsupported_shapes_are_reported(ta, seg):
    """malformed trees are refused (what is too deep for the flattened form cannot be expressed in it: the host
    mirror reports Unsupported for those, tests/test_host_cpu.py); nothing is silently approximated"""
    dev = ta.DeviceIndex([seg])
    try:
        # a conjunction that mixes nested occurs is malformed
        bad = (ta.MODE_BOOL, [1, 2, 3], [M, M, M], [0, 1, 1], 0, {"nested_occurs": [1, 1, 2], "atom_of": [0, 1, 1]})
        with pytest.raises(ta.TantivyAmdError):
            dev.search([bad], 10)
    finally:
        dev.close()


def test_nested_queries_on_a_small_segment_with_saturated_tfs(ta):
    """tf >= 255 (the byte-wide tf saturates: the packed value is read) inside nested queries"""
    rng = np.random.default_rng(6)
    md = 50_000
    lists = []
    for df in (20000, 9000, 4000, 15000, 700, 12000, 2500, 30000):
        pl = random_postings(rng, md, df, max_tf=4)
        pl = [(d, 300 + (d % 7) if i % 50 == 0 else tf) for i, (d, tf) in enumerate(pl)]
        lists.append(pl)
    seg = O.build_segment(md, lists, rng.integers(1, 400, size=md).tolist())
    specs = [(shape(rng.permutation(8).tolist()), msm) for shape, msm in SHAPES]
    dev = ta.DeviceIndex([seg])
    try:
        _check(ta, dev, seg, specs, 10)
    finally:
        dev.close()


# ---- phrases inside boolean queries (`+"a b" +c`; tree_kernel<KPL, true>)
@pytest.fixture(scope="module")
def pseg():
    return O.synth_segment(300_000, n_terms=48, with_positions=True, phrase_terms=16)


def test_phrases_inside_boolean_queries(ta, pseg):
    """every shape of tests/tree_shapes.PHRASE_SHAPES over frequent and rare lists: a PhraseQuery as a Must / Should /
    MustNot clause, as a member of a nested query, two phrases in one query; doc ids exact against the oracle (the C
    PhraseScorer under numpy's complex_scorer), scores within 1e-5, pruned == exhaustive, Count"""
    from tests.tree_shapes import PHRASE_SHAPES

    rng = np.random.default_rng(11)
    specs = []
    for shape, msm in PHRASE_SHAPES:
        for hi in (6, 12, 16):  # the most frequent lists (many matches), then rarer ones (probe tables + directories)
            specs.append((shape(rng.permutation(hi)[:8].tolist() if hi >= 8 else (rng.permutation(hi).tolist() + [6, 7])), msm))
    dev = ta.DeviceIndex([pseg])
    try:
        dev.set_option("dense_ratio", 16)
        n_hits = 0
        for k in (1, 10, 100):
            out = _check(ta, dev, pseg, specs, k)
            n_hits += int(out[3].sum())
        assert n_hits > 500
        counts = dev.count([to_device(ta, sp, msm) for sp, msm in specs])
        for i, (sp, msm) in enumerate(specs):
            assert int(counts[i]) == len(O.tree_match_all(pseg, to_oracle(sp), msm)[0]), (sp, msm)
        st = dev.segment_stats(0)
        assert st["n_dense_lists"] < 16  # some phrase lists were reached through probe tables
    finally:
        dev.close()


def test_phrases_inside_boolean_queries_with_deletes_and_large_tfs(ta):
    """term freqs >= 255 in phrase lists (the tf byte saturates: the packed values give the position index), a fifth
    of the docs deleted, one query per call from 8 threads"""
    rng = np.random.default_rng(12)
    docs = []
    for d in range(6000):
        n = int(rng.integers(3, 30))
        toks = rng.integers(0, 6, size=n).tolist()
        if d % 97 == 0:
            toks = [0, 1] * 300 + toks  # tf 300 of terms 0 and 1, the phrase "0 1" 300 times
        docs.append(" ".join("t%d" % t for t in toks))
    from tests.helpers import corpus_segment

    seg, vocab = corpus_segment(docs, with_positions=True)
    assert [vocab["t%d" % i] for i in range(6)] == list(range(6))
    deleted = np.sort(rng.choice(seg.max_doc, size=seg.max_doc // 5, replace=False)).astype(np.uint32)
    alive = np.ones(seg.max_doc, bool)
    alive[deleted] = False
    bits = np.packbits(alive, bitorder="little")
    body = np.uint32(seg.max_doc).tobytes() + bits.tobytes() + b"\0" * ((-len(bits)) % 8)
    from tests.tree_shapes import PHRASE_SHAPES

    specs = [(shape([0, 1, 2, 3, 4, 5, 0, 1]), msm) for shape, msm in PHRASE_SHAPES]
    specs += [(shape([1, 0, 3, 2, 5, 4, 1, 0]), msm) for shape, msm in PHRASE_SHAPES]
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_alive_bitset(body)
        _check(ta, dev, seg, specs, 10, deleted)
        queries = [to_device(ta, sp, msm) for sp, msm in specs]
        sc, _, dc, ct, _, _ = dev.search_concurrent(queries, 10, 8)
        for i, (sp, msm) in enumerate(specs):
            want = O.tree_search(seg, to_oracle(sp), 10, msm, deleted)
            assert sorted(int(dc[i, j]) for j in range(int(ct[i]))) == sorted(d for _, d in want), (sp, msm)
    finally:
        dev.close()


def test_phrase_in_boolean_errors(ta, pseg):
    """a one-term phrase is the reference's InvalidArgument; nine terms stay on the CPU (Unsupported; up to 8 run on
    the device since round 6: tests/test_gpu_round6.py)"""
    dev = ta.DeviceIndex([pseg])
    try:
        one = (ta.MODE_BOOL, [1, 2], [M, M], [0, 1], 0, {"nested_occurs": [M | 0x10, M], "atom_of": [0, 0], "phrase_offsets": [0, 0]})
        with pytest.raises(ta.TantivyAmdError):
            dev.search([one], 10)
        nine = (ta.MODE_BOOL, list(range(10)), [M] * 10, [0] * 9 + [1], 0,
                {"nested_occurs": [M | 0x10] * 9 + [M], "atom_of": [0] * 10, "phrase_offsets": list(range(9)) + [0]})
        with pytest.raises(ta.TantivyAmdError):
            dev.search([nine], 10)
    finally:
        dev.close()


def test_phrases_with_absent_terms(ta):
    """a phrase naming a term the segment does not hold is an EmptyScorer (Must: nothing matches; Should / MustNot: the
    clause is dropped), also one level down (boolean_weight.rs:255-257, 340-349)"""
    rng = np.random.default_rng(13)

    def docs_of(n, seed):
        r = np.random.default_rng(seed)
        out = []
        for _ in range(n):
            toks = r.integers(0, 5, size=int(r.integers(3, 25))).tolist()  # terms t0..t4; t5 never occurs
            out.append(" ".join("t%d" % t for t in toks) + " t6")
        return out

    from tests.helpers import corpus_segment

    seg, vocab = corpus_segment(docs_of(5000, 1), with_positions=True)
    ids = [vocab["t%d" % i] for i in range(5)]
    absent = len(vocab) + 3  # a term id the segment has no TermInfo for
    a, b, c, d = ids[0], ids[1], ids[2], ids[3]
    specs = [
        ([(M, ("ph", [a, absent])), (M, c)], 0),             # +"a ?" +c: nothing
        ([(S, ("ph", [a, absent])), (S, c)], 0),             # "a ?" c: c alone
        ([(M, c), (N, ("ph", [absent, b]))], 0),             # +c -"? b": c
        ([(M, a), (M, [(S, ("ph", [b, absent])), (S, d)], 0)], 0),  # +a +("b ?" d)
        ([(M, ("ph", [a, b])), (S, absent)], 0),             # +"a b" ?: the optional term is dropped
    ]
    dev = ta.DeviceIndex([seg])
    try:
        queries = [to_device(ta, sp, msm) for sp, msm in specs]
        sc, _, dc, ct = dev.search(queries, 10)
        assert int(ct[0]) == 0
        for i, (sp, msm) in enumerate(specs[1:], start=1):
            # the oracle's view of an absent term: drop it the way complex_scorer drops an EmptyScorer
            def strip(cl):
                if isinstance(cl, tuple) and len(cl) >= 2 and cl[0] == "ph":
                    return None if absent in cl[1] else cl
                return None if cl == absent else cl
            tree = []
            for cl in sp:
                if isinstance(cl[1], list):
                    members = [(o, strip(m)) for o, m in cl[1]]
                    members = [(o, m) for o, m in members if m is not None]
                    tree.append((cl[0], members, cl[2]))
                elif strip(cl[1]) is not None:
                    tree.append(cl)
            want = O.tree_search(seg, tree, 10, msm)
            got = [(float(sc[i, j]), int(dc[i, j])) for j in range(int(ct[i]))]
            assert sorted(x[1] for x in got) == sorted(x[1] for x in want), (sp, got, want)
            for (gs, _), (ws, _) in zip(sorted(got, key=lambda x: x[1]), sorted(want, key=lambda x: x[1])):
                assert rel_close(gs, ws, 1e-5), (sp, got, want)
    finally:
        dev.close()
