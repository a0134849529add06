"""The nested-tree oracle (oracle.tree_match_all: BooleanWeight::complex_scorer, boolean_weight.rs:236-431, restated
on every level in numpy) pinned against the pieces that are pinned themselves: on shapes without real nesting it must
equal oracle.bool_match_all (which the C scorer tree of to_query.c and the reference's KATs pin), bit for bit; on nested
shapes its doc sets must equal set algebra over the lists.  Also: the host-side flattening of a tree (tests/tree_shapes
.to_device) names every term once with consistent clause / member ids."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.tree_shapes import SHAPES, to_device

M, S, N = O.MUST, O.SHOULD, O.MUST_NOT


@pytest.fixture(scope="module")
def seg():
    return O.synth_segment(80_000, n_terms=32)


def _docs(seg, t):
    return set(O.decode_postings(seg, t)[0].tolist())


def test_flat_and_union_shapes_equal_the_two_level_oracle(seg):
    cases = [
        ([1, 2, 3], [M, S, N], None, 0, [(M, 1), (S, 2), (N, 3)]),
        ([4, 2, 0, 5], [M, M, M, M], [0, 1, 1, 1], 0, [(M, 4), (M, [(S, 2), (S, 0), (S, 5)], 0)]),
        ([3, 7, 9, 11], [M, M, M, M], [0, 0, 1, 1], 0, [(M, [(S, 3), (S, 7)], 0), (M, [(S, 9), (S, 11)], 0)]),
        ([1, 5, 9], [S, S, S], None, 2, [(S, 1), (S, 5), (S, 9)]),
        ([2, 6, 8, 10], [M, S, S, N], None, 1, [(M, 2), (S, 6), (S, 8), (N, 10)]),
    ]
    for terms, occ, cof, msm, tree in cases:
        d1, s1 = O.bool_match_all(seg, terms, occ, cof, msm)
        d2, s2 = O.tree_match_all(seg, tree, msm)
        assert np.array_equal(d1, d2), (terms, occ)
        assert np.array_equal(s1, s2), (terms, occ)
        assert len(d1) > 0


def test_nested_doc_sets_are_the_set_algebra_of_the_lists(seg):
    t = [0, 1, 2, 3, 4, 5, 6, 7]
    L = [_docs(seg, x) for x in t]
    want = [
        L[0] & ((L[1] & L[2]) | L[3]),
        (L[1] & L[2]) | L[3],
        L[0] - (L[1] & L[2]),
        L[0] | ((L[1]) - L[3]),
        L[0] & ((L[1] & L[2]) | (L[1] & L[3]) | (L[2] & L[3])),
        None,
        L[0] - (L[1] & L[2]),
        L[0] & ((L[3] | L[4]) - (L[1] & L[2])),
        (L[0] & L[1]) | (L[2] & L[3]) | (L[4] & L[5] & L[6]),
    ]
    for (shape, msm), w in zip(SHAPES, want):
        d, s = O.tree_match_all(seg, shape(t), msm)
        if w is None:  # (a b) (c d) (e f) ~2: at least two of the three unions
            u = [L[0] | L[1], L[2] | L[3], L[4] | L[5]]
            w = (u[0] & u[1]) | (u[0] & u[2]) | (u[1] & u[2])
        assert set(d.tolist()) == w, shape(t)
        assert (s > 0).all()


def test_scores_are_the_sum_of_the_matching_members(seg):
    """(+b +c) d: a doc holding b, c and d scores b + c + d; one holding only b and d scores d alone — the half
    matched intersection adds nothing"""
    b, c, d_ = 1, 2, 3
    docs, sc = O.tree_match_all(seg, [(S, [(M, b), (M, c)], 0), (S, d_)])
    per = {}
    for t in (b, c, d_):
        dd, ss = O.match_all(seg, [t], O.MODE_OR)
        per[t] = dict(zip(dd.tolist(), ss.tolist()))
    for doc, s in zip(docs.tolist()[:4000], sc.tolist()[:4000]):
        want = np.float32(0)
        if doc in per[b] and doc in per[c]:
            want = np.float32(np.float32(per[b][doc]) + np.float32(per[c][doc]))
        if doc in per[d_]:
            want = np.float32(want + np.float32(per[d_][doc]))
        assert abs(float(want) - s) <= 1e-5 * s


def test_device_tuples_name_every_term_once():
    import tantivy_amd as ta

    for shape, msm in SHAPES:
        spec = shape(list(range(10, 18)))
        q = to_device(ta, spec, msm)
        n = len(q[1])
        assert len(q[2]) == len(q[3]) == len(q[5]["nested_occurs"]) == len(q[5]["atom_of"]) == n
        flat = []
        for cl in spec:
            if isinstance(cl[1], (list, tuple)):
                for _, m in cl[1]:
                    flat += m if isinstance(m, (list, tuple)) else [m]
            else:
                flat.append(cl[1])
        assert q[1] == flat


# ---- phrases inside boolean queries (tests/tree_shapes.PHRASE_SHAPES)
@pytest.fixture(scope="module")
def pseg():
    return O.synth_segment(60_000, n_terms=16, with_positions=True, phrase_terms=8)


def _phrase_docs(seg, terms):
    d, s = O.match_all(seg, list(terms), O.MODE_PHRASE)
    return dict(zip(d.tolist(), s.tolist()))


def test_phrase_leaves_follow_the_set_algebra_and_score_like_the_phrase_scorer(pseg):
    """a PhraseQuery inside a BooleanQuery is one more scorer of complex_scorer: its doc set is the phrase's (the C
    restatement of PhraseScorer, pinned by the reference's phrase KATs), its score the phrase's score"""
    from tests.tree_shapes import PHRASE_SHAPES

    t = [0, 1, 2, 3, 4, 5]
    L = [_docs(pseg, x) for x in t]
    ab, abc, cd, bc = (_phrase_docs(pseg, x) for x in ([0, 1], [0, 1, 2], [2, 3], [1, 2]))
    assert len(ab) > 50 and len(bc) > 50
    A, ABC, CD, BC = set(ab), set(abc), set(cd), set(bc)
    n2 = lambda *sets: {d for d in set().union(*sets) if sum(d in x for x in sets) >= 2}
    want = [
        A & L[2],
        A,
        A | L[2],
        L[2] - A,
        (ABC & L[3]) - L[4],
        L[0] & (BC | L[3]),
        n2(A, CD, L[4]),
        (L[0] & L[1]) - CD,
        A & BC,
    ]
    for (shape, msm), w in zip(PHRASE_SHAPES, want):
        d, s = O.tree_match_all(pseg, shape(t), msm)
        assert set(d.tolist()) == w, shape(t)
        assert (s > 0).all()
    # +"a b" +c scores phrase + term, in either order bit for bit (two summands)
    d, s = O.tree_match_all(pseg, [(M, ("ph", [0, 1])), (M, 2)])
    dc, sc = O.match_all(pseg, [2], O.MODE_OR)
    c_score = dict(zip(dc.tolist(), sc.tolist()))
    for doc, got in zip(d.tolist(), s.tolist()):
        assert np.float32(got) == np.float32(np.float32(ab[doc]) + np.float32(c_score[doc]))


def test_device_tuples_of_phrase_shapes():
    import tantivy_amd as ta
    from tests.tree_shapes import PHRASE_SHAPES

    for shape, msm in PHRASE_SHAPES:
        spec = shape(list(range(8)))
        q = to_device(ta, spec, msm)
        ex = q[5]
        n = len(q[1])
        assert len(ex["nested_occurs"]) == len(ex["atom_of"]) == len(ex["phrase_offsets"]) == n
        # the terms of one phrase: same clause, same member, the flag on each, offsets 0..n-1
        groups = {}
        for i in range(n):
            if ex["nested_occurs"][i] & 0x10:
                groups.setdefault((q[3][i], ex["atom_of"][i]), []).append(ex["phrase_offsets"][i])
            else:
                assert ex["phrase_offsets"][i] == 0
        assert groups and all(v == list(range(len(v))) and len(v) >= 2 for v in groups.values())
