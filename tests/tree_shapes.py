"""Nested boolean query shapes shared by tests/test_gpu_tree.py and tests/test_tree_oracle_cpu.py.

A spec is a list of clauses: (outer occur, term id) for a term clause, or (outer occur, [(inner occur, term id |
[term ids of a nested intersection]), ...], nested minimum_number_should_match) for a nested BooleanQuery — the
structure O.tree_match_all takes.  to_device() flattens it into the host mirror's query tuple (tqh_query: occurs /
clause_of / nested_occurs / atom_of / clause_min_should, include/tantivy_amd_host.h).  A term id can be
("ph", [term ids]): a PhraseQuery, as a clause of its own or as a member of a nested query (PHRASE_SHAPES)."""
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


# round 6: a UNION one level further down (TQ_NESTED_ANY: `("any", [term ids])` as a member) — the depth-3 trees whose
# innermost level is not a conjunction — and the widest minimum the flattened form holds
DEEP_SHAPES = [
    # (+(b c) +d) e: a union inside an intersection inside a union
    (lambda t: [(S, [(M, ("any", [t[1], t[2]])), (M, t[3])], 0), (S, t[4])], 0),
    # +a +(+(b c) +d): ... inside an intersection
    (lambda t: [(M, t[0]), (M, [(M, ("any", [t[1], t[2]])), (M, t[3])], 0)], 0),
    # +a +(+(b c) -(d e)): a required and an excluded union one level down
    (lambda t: [(M, t[0]), (M, [(M, ("any", [t[1], t[2]])), (N, ("any", [t[3], t[4]]))], 0)], 0),
    # ((a b) (+c +d) e)~2 f: unions, an intersection and a term under a nested minimum
    (lambda t: [(S, [(S, ("any", [t[0], t[1]])), (S, [t[2], t[3]]), (S, t[4])], 2), (S, t[5])], 0),
    # +(+(a b) +(c d)) -(e f): two unions required together, a union excluded
    (lambda t: [(M, [(M, ("any", [t[0], t[1]])), (M, ("any", [t[2], t[3]]))], 0), (N, [(S, t[4]), (S, t[5])], 0)], 0),
]


def wide_minimum(t16, m):
    """16 Should terms, at least m of them (m <= 15): the bit-sliced counter of tq_tree.hip saturates at 15"""
    return [(M, [(S, x) for x in t16], m)]


# phrases inside boolean queries (`+"a b" +c`), over term ids with positions
PHRASE_SHAPES = [
    # +"a b" +c
    (lambda t: [(M, ("ph", [t[0], t[1]])), (M, t[2])], 0),
    # +"a b" c: the phrase required, a term optional (RequiredOptionalScorer)
    (lambda t: [(M, ("ph", [t[0], t[1]])), (S, t[2])], 0),
    # "a b" c: a union of a phrase and a term
    (lambda t: [(S, ("ph", [t[0], t[1]])), (S, t[2])], 0),
    # +c -"a b": a phrase under MustNot (Exclude)
    (lambda t: [(M, t[2]), (N, ("ph", [t[0], t[1]]))], 0),
    # +"a b c" +d -e: a three-term phrase
    (lambda t: [(M, ("ph", [t[0], t[1], t[2]])), (M, t[3]), (N, t[4])], 0),
    # +a +("b c" d): a phrase as a member of a nested union
    (lambda t: [(M, t[0]), (M, [(S, ("ph", [t[1], t[2]])), (S, t[3])], 0)], 0),
    # "a b" "c d" e ~2: two phrases and a term, at least two of them
    (lambda t: [(S, ("ph", [t[0], t[1]])), (S, ("ph", [t[2], t[3]])), (S, t[4])], 2),
    # +a +(+b -"c d"): a phrase excluded inside a nested query
    (lambda t: [(M, t[0]), (M, [(M, t[1]), (N, ("ph", [t[2], t[3]]))], 0)], 0),
    # +"a b" +"b c": two phrases that share a term
    (lambda t: [(M, ("ph", [t[0], t[1]])), (M, ("ph", [t[1], t[2]]))], 0),
]


def _is_phrase(x):
    return isinstance(x, tuple) and len(x) >= 2 and x[0] == "ph"


def to_oracle(spec):
    return spec


def to_device(ta, spec, msm=0):
    PH = 0x10  # TQ_NESTED_PHRASE
    terms, occurs, clause_of, nested, atom_of, offs, cms = [], [], [], [], [], [], {}
    any_phrase = False

    def add(t, outer, ci, inner, mi, off=0):
        terms.append(t)
        occurs.append(outer)
        clause_of.append(ci)
        nested.append(inner)
        atom_of.append(mi)
        offs.append(off)

    for ci, cl in enumerate(spec):
        if _is_phrase(cl[1]):  # a phrase as a clause of its own: a one-member group
            any_phrase = True
            for o, t in enumerate(cl[1][1]):
                add(t, cl[0], ci, M | PH, 0, o)
            continue
        if not isinstance(cl[1], (list, tuple)):
            add(cl[1], cl[0], ci, M, 0)
            continue
        for mi, (inner, member) in enumerate(cl[1]):
            if _is_phrase(member):
                any_phrase = True
                for o, t in enumerate(member[1]):
                    add(t, cl[0], ci, inner | PH, mi, o)
                continue
            if isinstance(member, tuple) and len(member) == 2 and member[0] == "any":  # a union one level down
                for t in member[1]:
                    add(t, cl[0], ci, inner | 0x20, mi)  # TQ_NESTED_ANY
                continue
            for t in (member if isinstance(member, (list, tuple)) else [member]):
                add(t, cl[0], ci, inner, mi)
        if len(cl) > 2 and cl[2]:
            cms[ci] = cl[2]
    extra = {"nested_occurs": nested, "atom_of": atom_of}
    if any_phrase:
        extra["phrase_offsets"] = offs
    if cms:
        extra["clause_min_should"] = cms
    return (ta.MODE_BOOL, terms, occurs, clause_of, msm, extra)
