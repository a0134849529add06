"""TermInfoStore: the oracle's restatement pinned by the reference's compat
fixtures (files written by released tantivy versions, tests/golden/compat_index.json) and by the
reference's own unit tests (term_info_store.rs:303-365); the product's C++ TermInfoStore /
TermInfoStoreWriter against the oracle.  Host-only code: no GPU needed."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compat_index.json")


def _fixture(ver):
    with open(GOLD) as f:
        j = json.load(f)[ver]
    return j, {k: bytes.fromhex(v) for k, v in j["files"].items()}


@pytest.mark.parametrize("ver", ["index_v6", "index_v7"])
def test_oracle_reads_the_reference_fixture(ver):
    """compat_tests.rs:39-57 opens these indexes and finds the one document with a term query;
    here: footer / composite framing, TermInfoStore header + block meta + TermInfo layout, and the
    TermInfo ranges must address posting lists that decode to doc_freq docs."""
    meta, files = _fixture(ver)
    assert meta["max_doc"] == 1
    term, _ = O.strip_footer(files["term"])
    idx, _ = O.strip_footer(files["idx"])
    pos, _ = O.strip_footer(files["pos"])
    fnorm, _ = O.strip_footer(files["fieldnorm"])
    terms_by_field = O.composite_fields(term)
    idx_by_field = O.composite_fields(idx)
    pos_by_field = O.composite_fields(pos)
    assert sorted(terms_by_field) == sorted(idx_by_field) == [(0, 0), (1, 0)]
    for key, record in (((0, 0), O.WITH_FREQS_AND_POSITIONS), ((1, 0), O.BASIC)):
        fst, store = O.term_dictionary_parts(terms_by_field[key])
        assert int.from_bytes(fst[:8], "little") in (1, 2, 3)   # tantivy-fst version header
        n = O.term_info_store_num_terms(store)
        assert n == 1                                           # one doc, one term per field
        df, ps, pe, qs, qe = O.term_info_store_get(store, 0)
        body = idx_by_field[key]
        assert df == 1 and ps == 0 and pe == len(body) - 8      # the list fills the sub-file
        if record == O.WITH_FREQS_AND_POSITIONS:
            assert (qs, qe) == (0, len(pos_by_field[key]))
        # the posting list decodes (vint tail of a 1-doc list) to doc 0, tf 1, position 0
        ti = O.TermInfo(df, ps, pe, qs, qe)
        seg = O.Segment(1, record, np.frombuffer(body, np.uint8),
                        np.frombuffer(pos_by_field.get(key, b""), np.uint8), None, [ti],
                        int.from_bytes(body[:8], "little"))
        docs, tfs = O.decode_postings(seg, 0)
        assert docs.tolist() == [0] and tfs.tolist() == [1]
        if record == O.WITH_FREQS_AND_POSITIONS:
            p, n_pos = O.decode_positions(seg, 0, 4)
            assert n_pos == 1 and p.tolist() == [0]
        # re-serialising the decoded TermInfo gives the fixture's store bytes back
        assert O.term_info_store_serialize([(df, ps, pe, qs, qe)]) == store
    assert len(O.composite_fields(fnorm)) >= 1


def _offset(i):
    return i * 13 + i * i


def _pack_infos(n):
    """term_info_store.rs:341-365 (test_pack)"""
    return [(i, _offset(i), _offset(i + 1), _offset(i) * 3, _offset(i + 1) * 3) for i in range(n)]


def test_oracle_test_pack_and_bitpacked():
    infos = _pack_infos(1000)
    store = O.term_info_store_serialize(infos)
    assert O.term_info_store_num_terms(store) == 1000
    for i in range(1000):
        assert O.term_info_store_get(store, i) == infos[i], i
    # term_info_store.rs:303-318 (test_bitpacked): widths of 321, 2, 51
    assert [O.compute_num_bits(v) for v in (321, 2, 51, 0, 1 << 56, (1 << 56) - 1)] == [9, 2, 6, 0, 64, 56]


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def _random_infos(rng, n, big=False):
    # list lengths are u32 in a TermInfo (posting_num_bytes); the offsets themselves are u64
    ps = (np.cumsum(rng.integers(1, (1 << 31) if big else 5000, size=n + 1)) + ((1 << 40) if big else 0)).tolist()
    qs = np.cumsum(rng.integers(0, (1 << 20) if big else 9000, size=n + 1)).tolist()
    df = rng.integers(1, (1 << 31) if big else 100000, size=n).tolist()
    return [(int(df[i]), int(ps[i]), int(ps[i + 1]), int(qs[i]), int(qs[i + 1])) for i in range(n)]


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1000, 2049])
def test_product_store_equals_oracle(ta, n):
    rng = np.random.default_rng(n)
    for infos in (_pack_infos(n), _random_infos(rng, n), _random_infos(rng, n, big=True)):
        want = O.term_info_store_serialize(infos)
        got = ta.TermInfoStore.serialize(infos)
        assert got == want
        st = ta.TermInfoStore(want)
        try:
            assert st.num_terms() == n
            assert st.get(list(range(n))) == infos
            probe = rng.integers(0, n, size=min(n, 64)).tolist()
            assert st.get(probe) == [O.term_info_store_get(want, i) for i in probe]
            with pytest.raises(ta.TantivyAmdError):
                st.get([n])
        finally:
            st.close()


def test_product_reads_the_reference_fixture(ta):
    _, files = _fixture("index_v7")
    term, _ = O.strip_footer(files["term"])
    for key, field_file in O.composite_fields(term).items():
        off, ln = ta.term_dictionary_values(field_file)
        _, store = O.term_dictionary_parts(field_file)
        assert field_file[off:off + ln] == store
        st = ta.TermInfoStore(store)
        try:
            assert st.get([0]) == [O.term_info_store_get(store, 0)]
        finally:
            st.close()
    with pytest.raises(ta.TantivyAmdError):
        ta.TermInfoStore(b"\x00" * 8)
    with pytest.raises(ta.TantivyAmdError):
        ta.term_dictionary_values(b"\x01")
