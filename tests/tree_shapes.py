"""Nested boolean query shapes shared by tests/test_gpu_tree.py and tests/test_tree_oracle_cpu.py.

A spec is a list of clauses: (outer occur, term id) for a term clause, or (outer occur, [(inner occur, term id |
[term ids of a nested intersection]), ...], nested minimum_number_should_match) for a nested BooleanQuery — the
structure O.tree_match_all takes.  to_device() flattens it into the host mirror's query tuple (tqh_query: occurs /
clause_of / nested_occurs / atom_of / clause_min_should, include/tantivy_amd_host.h)."""
from oracle import oracle as O

M, S, N = O.MUST, O.SHOULD, O.MUST_NOT

# (spec builder over 8 distinct term ids, top-level minimum_number_should_match)
SHAPES = [
    # +a +((+b +c) d): an intersection inside a union inside an intersection (`+a +(b AND c | d)`)
    (lambda t: [(M, t[0]), (M, [(S, [t[1], t[2]]), (S, t[3])], 0)], 0),
    # (+b +c) d: an intersection inside a union
    (lambda t: [(S, [(M, t[1]), (M, t[2])], 0), (S, t[3])], 0),
    # +a -(+b +c): an intersection under MustNot
    (lambda t: [(M, t[0]), (N, [(M, t[1]), (M, t[2])], 0)], 0),
    # a (+b c -d): a nested query with Must, optional and MustNot terms as a Should clause
    (lambda t: [(S, t[0]), (S, [(M, t[1]), (S, t[2]), (N, t[3])], 0)], 0),
    # +a +(b c d)~2: minimum_number_should_match inside the nested query
    (lambda t: [(M, t[0]), (M, [(S, t[1]), (S, t[2]), (S, t[3])], 2)], 0),
    # (a b) (c d) (e f) ~2: minimum_number_should_match over nested groups (disjunction.rs:113-139)
    (lambda t: [(S, [(S, t[0]), (S, t[1])], 0), (S, [(S, t[2]), (S, t[3])], 0), (S, [(S, t[4]), (S, t[5])], 0)], 2),
    # +a -(b c)~2 d: a nested minimum under MustNot, next to an optional term
    (lambda t: [(M, t[0]), (N, [(S, t[1]), (S, t[2])], 2), (S, t[3])], 0),
    # +a +(-(+b +c) d e): an excluded intersection one level further down
    (lambda t: [(M, t[0]), (M, [(N, [t[1], t[2]]), (S, t[3]), (S, t[4])], 0)], 0),
    # (+a +b) (+c +d) (+e +f +g): a union of intersections
    (lambda t: [(S, [(M, t[0]), (M, t[1])], 0), (S, [(M, t[2]), (M, t[3])], 0), (S, [(M, t[4]), (M, t[5]), (M, t[6])], 0)], 0),
]


def to_oracle(spec):
    return spec


def to_device(ta, spec, msm=0):
    terms, occurs, clause_of, nested, atom_of, cms = [], [], [], [], [], {}
    for ci, cl in enumerate(spec):
        if not isinstance(cl[1], (list, tuple)):
            terms.append(cl[1])
            occurs.append(cl[0])
            clause_of.append(ci)
            nested.append(M)
            atom_of.append(0)
            continue
        for mi, (inner, member) in enumerate(cl[1]):
            for t in (member if isinstance(member, (list, tuple)) else [member]):
                terms.append(t)
                occurs.append(cl[0])
                clause_of.append(ci)
                nested.append(inner)
                atom_of.append(mi)
        if len(cl) > 2 and cl[2]:
            cms[ci] = cl[2]
    extra = {"nested_occurs": nested, "atom_of": atom_of}
    if cms:
        extra["clause_min_should"] = cms
    return (ta.MODE_BOOL, terms, occurs, clause_of, msm, extra)
