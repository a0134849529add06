"""GPU tests added in round 6 (closing SURVEY §8 f1).

* Nested boolean queries whose innermost level is a UNION (tq_query.nested_occurs | TQ_NESTED_ANY: `(+(b c) +d) e`),
  16 Should inputs under a minimum (the bit-sliced counter saturates), phrases of up to 8 terms inside a boolean query.
* The doc sets the reference's own tests assert for boolean queries (boolean_query/mod.rs:109-219,
  boolean_query.rs:287-352), on the device — 5-doc corpora: lists below any density threshold get their bitmaps from
  the probe pool whatever the segment's size.
* 10M docs x 65 536 terms with "probe_budget_x" = 2: every shape of tests/tree_shapes.py over lists nobody has
  named before, batch after batch — the probe pool gives the slots of the lists used longest ago to the new ones
  (tq_segment_stats.probe_evictions) and no query is refused.
Oracle: O.tree_match_all (complex_scorer in numpy), which equals the C transliteration of the scorer tree on every
shape (tests/test_tree_oracle_cpu.py)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import corpus_segment, rel_close
from tests.tree_shapes import DEEP_SHAPES, PHRASE_SHAPES, SHAPES, to_device, to_oracle, wide_minimum

pytestmark = pytest.mark.gpu

M, S, N = O.MUST, O.SHOULD, O.MUST_NOT


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def _check(ta, dev, seg, specs, k, want_tree=True):
    queries = [to_device(ta, sp, msm) for sp, msm in specs]
    out = {}
    for mode in (0, 1):
        dev.set_option("exhaustive", mode)
        out[mode] = dev.search(queries, k)
        if want_tree:
            assert dev.last_batch_stats()["kernel_mask"] & ta.binding.KERNEL_TREE
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    sc, _, dc, ct = out[0]
    for i, (sp, msm) in enumerate(specs):
        want = O.tree_search(seg, to_oracle(sp), k, msm)
        got = [(float(sc[i, j]), int(dc[i, j])) for j in range(int(ct[i]))]
        assert len(got) == len(want), (sp, msm, got, want)
        if [d for _, d in got] != [d for _, d in want]:  # near-ties of 3+ term sums may swap neighbours
            assert sorted(d for _, d in got) == sorted(d for _, d in want), (sp, msm, got, want)
        for (gs, _), (ws, _) in zip(sorted(got, key=lambda x: x[1]), sorted(want, key=lambda x: x[1])):
            assert rel_close(gs, ws, 1e-5), (sp, msm, got, want)
    return out[0]


def test_unions_one_level_down_and_wide_minimums(ta):
    seg = O.synth_segment(300_000, n_terms=48)
    rng = np.random.default_rng(17)
    specs = []
    for shape, msm in DEEP_SHAPES:
        for _ in range(4):
            specs.append((shape(rng.permutation(40)[:8].tolist()), msm))
    for m in (2, 3, 9, 15):  # 16 Should terms: docs that hold all 16 must stay in (the counter saturates at 15)
        specs.append((wide_minimum(list(range(16)), m), 0))
        specs.append((wide_minimum(rng.permutation(24)[:16].tolist(), m), 0))
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 32)
        n = 0
        for k in (1, 10, 100):
            n += int(_check(ta, dev, seg, specs, k)[3].sum())
        assert n > 2000
        counts = dev.count([to_device(ta, sp, msm) for sp, msm in specs])
        for i, (sp, msm) in enumerate(specs):
            assert int(counts[i]) == len(O.tree_match_all(seg, to_oracle(sp), msm)[0]), (sp, msm)
    finally:
        dev.close()
    # docs that hold ALL 16 Should terms (a 4-plane counter without saturation wrapped to 0 and dropped them)
    md = 30_000
    lists = []
    for i in range(16):
        docs = sorted(set(range(0, md, 97)) | set(range(i, md, 5 + i)))
        lists.append([(d, 1 + (d + i) % 3) for d in docs])
    seg16 = O.build_segment(md, lists, [5 + d % 40 for d in range(md)])
    dev = ta.DeviceIndex([seg16])
    try:
        specs = [(wide_minimum(list(range(16)), m), 0) for m in (2, 8, 15)]
        out = _check(ta, dev, seg16, specs, 100)
        all16 = set(range(0, md, 97))
        for i in range(len(specs)):
            got = set(int(x) for x in out[2][i, :int(out[3][i])])
            assert got & all16, specs[i]  # (they score highest: 16 terms each)
        counts = dev.count([to_device(ta, sp, msm) for sp, msm in specs])
        for i, (sp, msm) in enumerate(specs):
            d, _ = O.tree_match_all(seg16, to_oracle(sp), msm)
            assert int(counts[i]) == len(d) and all16 <= set(d.tolist()), (sp, msm)
    finally:
        dev.close()


def test_reference_boolean_doc_set_kats_on_the_device(ta):
    """boolean_query/mod.rs:109-219, :48-56, boolean_query.rs:287-352 through the nested-query path of the device (a
    clause written as a one-member nested query takes tq_tree.hip) and through the flat boolean path"""
    seg, v = corpus_segment(["a b c", "a c", "b c", "a b c d", "d"], with_positions=False)
    a, b, c, d = (v[x] for x in "abcd")
    dev = ta.DeviceIndex([seg])
    try:
        def docs(spec, msm=0, k=10, tree=False):
            out = _check(ta, dev, seg, [(spec, msm)], k, want_tree=tree)
            return sorted(int(x) for x in out[2][0, :int(out[3][0])])

        # (one-member nested queries flatten into the boolean kernels; the others take tq_tree.hip)
        assert docs([(M, [(M, a)], 0), (S, [(M, b)], 0), (N, [(M, d)], 0)]) == [0, 1]
        assert docs([(S, [(M, a), (M, b)], 0), (S, d)], tree=True) == [0, 3, 4]   # `(+a +b) d`: count 3
        assert docs([(M, d), (N, [(M, a)], 0), (N, [(M, b)], 0)]) == [4]          # two excluded
        assert docs([(M, [(S, a), (S, b)], 0), (N, [(M, d)], 0)]) == [0, 1, 2]
        assert docs([(M, a), (N, [(M, b), (M, d)], 0)], tree=True) == [0, 1]      # an excluded intersection
        assert docs([(M, c), (M, [(M, ("any", [a, d])), (M, b)], 0)], tree=True) == [0, 3]  # `+c +(+(a d) +b)`
        seg2, v2 = corpus_segment(["a b c", "a c e", "d f g", "z z z", "c i b"], with_positions=False)
        dev2 = ta.DeviceIndex([seg2])
        try:
            def docs2(words, mr):
                spec = [(M, [(S, v2[w]) for w in words], mr)]
                out = _check(ta, dev2, seg2, [(spec, 0)], 10, want_tree=mr >= 2)
                return sorted(int(x) for x in out[2][0, :int(out[3][0])])

            assert docs2(["a", "c", "z", "i"], 2) == [0, 1, 4]
            assert docs2(["a", "b", "c", "e"], 3) == [0, 1]
            assert docs2(["a", "b"], 3) == []
        finally:
            dev2.close()
    finally:
        dev.close()


def test_eight_term_phrase_inside_a_boolean_query(ta):
    docs = ["a b c d e f g h x", "a b c d e f g h", "h g f e d c b a x", "x a b c d e f g h y a b c d e f g h", "a b c d x e f g h"] * 40
    docs += ["x y", "a x", "b c d"] * 30
    seg, v = corpus_segment(docs)
    ph = ("ph", [v[w] for w in "abcdefgh"])
    specs = [([(M, ph), (M, v["x"])], 0), ([(S, ph), (S, v["y"])], 0), ([(M, v["x"]), (N, ph)], 0),
             ([(M, v["a"]), (M, [(S, ph), (S, v["y"])], 0)], 0)]
    dev = ta.DeviceIndex([seg])
    try:
        out = _check(ta, dev, seg, specs, 10)
        assert int(out[3].sum()) >= 30
        counts = dev.count([to_device(ta, sp, msm) for sp, msm in specs])
        for i, (sp, msm) in enumerate(specs):
            assert int(counts[i]) == len(O.tree_match_all(seg, to_oracle(sp), msm)[0]), sp
    finally:
        dev.close()


def test_every_nested_shape_at_10m_docs_and_65536_terms_with_a_small_probe_budget(ta):
    """no tree is refused for budget reasons: the probe pool (probe_budget_x = 2: a few dozen slots) is far smaller
    than the lists the batches name; every batch runs on tq_tree.hip, >= 32 queries per shape against the oracle"""
    vocab = 65536
    seg = O.synth_segment(10_000_000, n_terms=vocab)
    rng = np.random.default_rng(23)
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("probe_budget_x", 2)
        shapes = SHAPES + DEEP_SHAPES
        checked = {i: 0 for i in range(len(shapes))}
        for batch in range(4):
            specs, owner = [], []
            for si, (shape, msm) in enumerate(shapes):
                for _ in range(8):
                    # ranks spread over the whole vocabulary: mostly lists without tables of their own
                    ids = np.unique(np.concatenate([rng.integers(0, 64, size=2), (vocab ** rng.random(10)).astype(np.int64) - 1]))
                    ids = rng.permutation(ids)[:8].tolist()
                    if len(ids) < 8:
                        continue
                    specs.append((shape(ids), msm))
                    owner.append(si)
            _check(ta, dev, seg, specs, 10)
            for si in owner:
                checked[si] += 1
        assert min(checked.values()) >= 30, checked
        st = dev.segment_stats(0)
        assert st["probe_evictions"] > 0, st
    finally:
        dev.close()
