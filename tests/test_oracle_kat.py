"""Pins the CPU ORACLE against the reference's own known-answer tests.
Each test cites the reference test it replays.  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import corpus_segment, nearly_equals, random_postings

L = O.lib()
TERMINATED = O.TERMINATED


# ------------------------------------------------------------------ fieldnorm (fieldnorm/code.rs:277-328)
def test_fieldnorm_table_and_inverse():
    t = O.fieldnorm_table()
    assert len(t) == 256
    assert list(t[:41]) == list(range(41))
    assert int(t[255]) == 2_013_265_944
    assert list(t[251:]) == [1_476_395_032, 1_610_612_760, 1_744_830_488, 1_879_048_216,
                             2_013_265_944]
    for i in range(41):
        assert O.fieldnorm_to_id(i) == i
    assert O.fieldnorm_to_id(41) == 40
    assert O.fieldnorm_to_id(42) == 41
    for i in range(43, 256):
        fnv = int(t[i])
        assert O.fieldnorm_to_id(fnv) == i
        assert O.fieldnorm_to_id(fnv - 1) == i - 1
        assert O.fieldnorm_to_id(fnv + 1) == i
    assert O.fieldnorm_to_id(0xFFFFFFFF) == 255
    # fieldnorm/reader.rs:168-193
    assert int(t[O.fieldnorm_to_id(300)]) == 280
    assert int(t[O.fieldnorm_to_id(1_000_000)]) == 983_064


# ------------------------------------------------------------------ BM25 (bm25.rs:235-239)
def test_idf():
    assert nearly_equals(L.to_idf(1, 2), math.log(2.0))
    assert L.to_idf(1, 2) == np.float32(np.log(np.float32(2.0)))


# ------------------------------------------------------------------ vint (compression/mod.rs:359-376, compat hex)
def test_vint_wire_examples():
    assert bytes(O.vint_compress_sorted([0], 0)) == b"\x80"
    assert bytes(O.vint_compress_unsorted([1])) == b"\x81"
    buf = (C.c_uint8 * 10)()
    n = L.to_vint_serialize(5, buf)
    assert bytes(buf[:n]) == b"\x85"
    n = L.to_vint_serialize(300, buf)
    assert bytes(buf[:n]) == bytes([300 % 128, (300 // 128) | 0x80])
    v = C.c_uint64()
    assert L.to_vint_deserialize(buf, n, C.byref(v)) == 2 and v.value == 300


@pytest.mark.parametrize("offset", [0, 1, 2])
def test_vint_block_roundtrip(offset):
    # test_encode_vint: input 4 + i*7/2 for i<123, <= 154 bytes, padding preserved
    vals = np.array([4 + i * 7 // 2 for i in range(123)], dtype=np.uint32)
    enc = O.vint_compress_sorted(vals, offset)
    assert len(enc) <= 154
    padded = np.concatenate([enc, np.full(7, 0xAB, np.uint8)])
    c, out = O.vint_uncompress_sorted(padded, 123, offset)
    assert c == len(enc)
    assert np.array_equal(out, vals)
    enc2 = O.vint_compress_unsorted(vals)
    c2, out2 = O.vint_uncompress_unsorted(enc2, 123)
    assert c2 == len(enc2) and np.array_equal(out2, vals)


# ------------------------------------------------------------------ block codec (compression/mod.rs:277-351)
def test_encode_sorted_block():
    vals = np.arange(128, dtype=np.uint32) * 7
    nb, data = O.compress_block_sorted(vals, 0)
    assert len(data) == nb * 16
    n, out = O.uncompress_block_sorted(data, 0, nb)
    assert n == len(data) and np.array_equal(out, vals)


def test_encode_sorted_block_with_offset_and_junk():
    vals = 11 + np.arange(128, dtype=np.uint32) * 7
    nb, data = O.compress_block_sorted(vals, 10)
    junk = np.concatenate([data, np.array([173], np.uint8)])
    n, out = O.uncompress_block_sorted(junk, 10, nb)
    assert n == len(data) == nb * 16
    assert np.array_equal(out, vals)


def test_encode_unsorted_block_with_junk():
    vals = (np.arange(128, dtype=np.uint32) * 7) % 12
    for minus_one, v in ((False, vals), (True, vals + 1)):
        nb, data = O.compress_block_unsorted(v, minus_one)
        junk = np.concatenate([data, np.array([173], np.uint8)])
        n, out = O.uncompress_block_unsorted(junk, nb, minus_one)
        assert n == len(data) == nb * 16
        assert np.array_equal(out, v)


def test_block_all_bit_widths_roundtrip():
    rng = np.random.default_rng(7)
    for b in range(0, 33):
        hi = (1 << b) if b < 32 else (1 << 32)
        vals = rng.integers(0, hi, size=128, dtype=np.uint64).astype(np.uint32)
        if b:
            vals[5] = hi - 1
        nb, data = O.compress_block_unsorted(vals, False)
        assert nb == b and len(data) == 16 * b
        n, out = O.uncompress_block_unsorted(data, nb, False)
        assert np.array_equal(out, vals)
    for b in range(0, 32):
        gaps = rng.integers(0, (1 << b) if b else 1, size=128, dtype=np.uint64)
        if b:
            gaps[3] = (1 << b) - 1
        if (int(gaps.sum()) + 200) >= TERMINATED:
            gaps = gaps // 256
        for offset in (0, 77):
            vals = (offset + np.cumsum(gaps + 1) - (1 if offset == 0 else 0)).astype(np.uint32)
            nb, data = O.compress_block_sorted(vals, offset)
            assert len(data) == nb * 16
            n, out = O.uncompress_block_sorted(data, offset, nb)
            assert np.array_equal(out, vals)


def test_first_block_stores_first_value_raw():
    # SURVEY §A.1 / compression/mod.rs:36-39: offset 0 <-> None, a block may start at doc 0
    vals = np.arange(128, dtype=np.uint32)
    nb, data = O.compress_block_sorted(vals, 0)
    assert nb == 0 and len(data) == 0
    n, out = O.uncompress_block_sorted(data, 0, 0)
    assert np.array_equal(out, vals)


def test_search_block_matches_linear_scan():
    # block_search.rs:88-179 (proptest vs linear)
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(0, 129))
        docs = np.sort(rng.choice(5000, size=n, replace=False)).astype(np.uint32)
        arr = np.full(128, TERMINATED, np.uint32)
        arr[:n] = docs
        for target in list(docs[:5]) + [0, 1, 4999, 5001, int(rng.integers(0, 5001))]:
            expect = int(np.searchsorted(arr, target, side="left"))
            got = L.to_search_block(arr.ctypes.data_as(C.POINTER(C.c_uint32)), int(target))
            assert got == expect


# ------------------------------------------------------------------ skip (skip.rs:314-462)
def test_skip_codes():
    assert L.to_encode_bitwidth(2, 1) == 0b01000010
    assert L.to_encode_bitwidth(2, 0) == 0b00000010
    for tf in range(255):
        assert L.to_encode_block_wand_max_tf(tf) == tf
        assert L.to_decode_block_wand_max_tf(tf) == tf
    for tf in (255, 256, 1_000_000, 0xFFFFFFFF):
        assert L.to_encode_block_wand_max_tf(tf) == 255
    assert L.to_decode_block_wand_max_tf(255) == 0xFFFFFFFF


class SkipState(C.Structure):
    _fields_ = [("last_doc_in_block", C.c_uint32), ("is_vint", C.c_int), ("doc_num_bits", C.c_uint8),
                ("strict", C.c_int), ("tf_num_bits", C.c_uint8), ("tf_sum", C.c_uint32),
                ("bw_fieldnorm_id", C.c_uint8), ("bw_term_freq", C.c_uint32),
                ("num_docs", C.c_uint32), ("byte_offset", C.c_uint64),
                ("position_offset", C.c_uint64)]


def _skip_walk(buf, doc_freq, skip_info, n_adv):
    L.to_skip_walk.restype = C.c_size_t
    L.to_skip_walk.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.c_size_t,
                               C.POINTER(SkipState)]
    out = (SkipState * (n_adv + 1))()
    L.to_skip_walk(bytes(buf), len(buf), doc_freq, skip_info, n_adv, out)
    return list(out)


def _skip_entry(last_doc, doc_bits, tf_bits=None, fn=None, tf=None):
    b = int(last_doc).to_bytes(4, "little") + bytes([L.to_encode_bitwidth(doc_bits, 1)])
    if tf_bits is not None:
        b += bytes([tf_bits, fn, L.to_encode_block_wand_max_tf(tf)])
    return b


def test_skip_with_freq():
    buf = _skip_entry(1, 2, 3, 13, 3) + _skip_entry(5, 5, 2, 8, 2)
    st = _skip_walk(buf, 3 + 256, O.WITH_FREQS, 4)
    s = st[0]
    assert (s.last_doc_in_block, s.is_vint, s.doc_num_bits, s.strict, s.tf_num_bits, s.tf_sum,
            s.bw_fieldnorm_id, s.bw_term_freq) == (1, 0, 2, 1, 3, 0, 13, 3)
    s = st[1]
    assert (s.last_doc_in_block, s.is_vint, s.doc_num_bits, s.strict, s.tf_num_bits, s.tf_sum,
            s.bw_fieldnorm_id, s.bw_term_freq) == (5, 0, 5, 1, 2, 0, 8, 2)
    assert (st[2].is_vint, st[2].num_docs) == (1, 3)
    assert st[2].byte_offset == 16 * (2 + 3) + 16 * (5 + 2)
    assert (st[3].is_vint, st[3].num_docs) == (1, 0)
    assert (st[4].is_vint, st[4].num_docs) == (1, 0)
    assert st[3].last_doc_in_block == TERMINATED


def test_skip_no_freq_and_multiple_of_block_size():
    buf = _skip_entry(1, 2) + _skip_entry(5, 5)
    st = _skip_walk(buf, 3 + 256, O.BASIC, 4)
    assert (st[0].last_doc_in_block, st[0].doc_num_bits, st[0].tf_num_bits) == (1, 2, 0)
    assert (st[1].last_doc_in_block, st[1].doc_num_bits) == (5, 5)
    assert (st[2].is_vint, st[2].num_docs) == (1, 3)
    assert (st[3].is_vint, st[3].num_docs) == (1, 0)
    st = _skip_walk(_skip_entry(1, 2), 128, O.BASIC, 1)
    assert (st[0].last_doc_in_block, st[0].is_vint, st[0].doc_num_bits) == (1, 0, 2)
    assert (st[1].is_vint, st[1].num_docs) == (1, 0)


# ------------------------------------------------------------------ TermScorer KATs (term_scorer.rs:168-278)
class TS:
    def __init__(self, seg, term, w):
        L.to_ts_new.restype = C.c_void_p
        L.to_ts_new.argtypes = [C.POINTER(O.SegmentView), C.POINTER(O.TermInfo), C.POINTER(O.Bm25)]
        for f in ("to_ts_doc", "to_ts_advance", "to_ts_term_freq", "to_ts_last_doc_in_block"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.to_ts_seek.restype = C.c_uint32
        L.to_ts_seek.argtypes = [C.c_void_p, C.c_uint32]
        L.to_ts_seek_block.argtypes = [C.c_void_p, C.c_uint32]
        for f in ("to_ts_score", "to_ts_block_max_score", "to_ts_max_score"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_void_p]
        L.to_ts_free.argtypes = [C.c_void_p]
        self.seg, self.w = seg, w
        self.h = L.to_ts_new(C.byref(seg.view), C.byref(seg.terms[term]), C.byref(w))
        assert self.h

    def __getattr__(self, name):
        fn = getattr(L, "to_ts_" + name)
        return lambda *a: fn(self.h, *a)


def test_term_scorer_max_score():
    seg = O.build_segment(8, [[(2, 3), (3, 12), (7, 8)]], [0, 0, 10, 12, 0, 0, 0, 100])
    w = O.bm25_for_one_term(3, 6, 10.0)
    ts = TS(seg, 0, w)
    assert nearly_equals(ts.max_score(), 1.3990127)
    assert ts.doc() == 2 and ts.term_freq() == 3
    assert nearly_equals(ts.block_max_score(), 1.3676447)
    assert nearly_equals(ts.score(), 1.0892314)
    assert ts.advance() == 3 and ts.term_freq() == 12
    assert nearly_equals(ts.score(), 1.3676447)
    assert ts.advance() == 7 and ts.term_freq() == 8
    assert nearly_equals(ts.score(), 0.72015285)
    assert ts.advance() == TERMINATED
    # exact f32 (SURVEY: numpy-f32 restatement reproduces all printed digits)
    ts2 = TS(seg, 0, w)
    assert np.float32(ts2.score()) == np.float32(1.0892314)
    assert np.float32(ts2.max_score()) == np.float32(1.3990127)


def test_term_scorer_shallow_advance():
    doc_tfs = [(i * 10, 1 + (i * 10) % 3) for i in range(300)]
    seg = O.build_segment(3000, [doc_tfs], [10] * 3000)
    ts = TS(seg, 0, O.bm25_for_one_term(300, 1024, 10.0))
    assert ts.doc() == 0
    ts.seek_block(1289)
    assert ts.doc() == 0
    assert ts.seek(1289) == 1290


def test_block_wand_block_max_kat():
    doc_tfs = [(d, 1) for d in range(128)]
    doc_tfs += [(d, 2 if d == 200 else 1) for d in range(128, 256)]
    doc_tfs += [(256, 1), (257, 3), (258, 1)]
    seg = O.build_segment(300, [doc_tfs], [20] * 300)
    ts = TS(seg, 0, O.bm25_for_one_term(10, 129, 20.0))
    assert nearly_equals(ts.block_max_score(), 2.5161593)
    ts.seek_block(135)
    assert nearly_equals(ts.block_max_score(), 3.4597192)
    ts.seek_block(256)
    assert nearly_equals(ts.block_max_score(), 5.2971773)  # unloaded vint tail => max_score()
    assert ts.seek(256) == 256
    assert nearly_equals(ts.block_max_score(), 3.9539647)
    for got, want in ((np.float32(ts.block_max_score()), np.float32(3.9539647)),):
        assert got == want


def test_term_scorer_block_max_equals_max_in_block():
    # proptest term_scorer.rs:211-252
    rng = np.random.default_rng(11)
    for _ in range(25):
        n = int(rng.integers(80, 300))
        tfs = rng.integers(1, 10, size=n)
        extra = rng.integers(0, 100, size=n)
        fieldnorms = (tfs + extra).tolist()
        doc_tfs = [(d, int(tfs[d])) for d in range(n)]
        seg = O.build_segment(n, [doc_tfs], fieldnorms)
        avg = float(np.float32(sum(fieldnorms)) / np.float32(n))
        ts = TS(seg, 0, O.bm25_for_one_term(n, n * 10, avg))
        for start in range(0, n, 128):
            bm = ts.block_max_score()
            best = 0.0
            for d in range(start, min(n, start + 128)):
                assert ts.doc() == d
                best = max(best, ts.score())
                ts.advance()
            assert nearly_equals(best, bm)


# ------------------------------------------------------------------ positions (postings/mod.rs:60-82)
def test_position_write_size():
    plist = [(d, 4) for d in range(120)]
    pos = [[1, 3, 6, 8] for _ in range(120)]  # deltas 1,2,3,2
    seg = O.build_segment(120, [plist], [4] * 120, record_option=O.WITH_FREQS_AND_POSITIONS,
                          positions=[pos], total_num_tokens=480)
    # 1 (VInt 3 blocks) + 3 width bytes + 3*32 bitpacked (2 bits) + 96 vint; the reference's 207
    # adds the 11-byte CompositeFile footer, which is outside the hot path.
    assert seg.pos_len == 196 == 207 - 11
    flat, n = O.decode_positions(seg, 0, 480)
    assert n == 480 and list(flat[:8]) == [1, 3, 6, 8, 1, 3, 6, 8]


def test_positions_roundtrip_random():
    rng = np.random.default_rng(5)
    plist, pos = [], []
    for d in range(0, 900, 2):
        tf = int(rng.integers(1, 30))
        ps = np.sort(rng.choice(5000, size=tf, replace=False)).tolist()
        plist.append((d, tf))
        pos.append(ps)
    seg = O.build_segment(900, [plist], [5000] * 900, record_option=O.WITH_FREQS_AND_POSITIONS,
                          positions=[pos])
    total = sum(tf for _, tf in plist)
    flat, n = O.decode_positions(seg, 0, total)
    assert n == total
    assert flat.tolist() == [p for ps in pos for p in ps]


# ------------------------------------------------------------------ seek / boundaries (block_segment_postings.rs:467-631)
def _basic_seg(docs):
    return O.build_segment(max(docs) + 1, [[(d, 1) for d in docs]], None, record_option=O.BASIC)


def test_consecutive_docs_roundtrip():
    docs = list(range(100_000))
    seg = _basic_seg(docs)
    got, _ = O.decode_postings(seg, 0)
    assert got.tolist() == docs


def test_skip_right_at_new_block():
    docs = list(range(128)) + [129, 130]
    seg = _basic_seg(docs)
    w = O.bm25_for_one_term(1, 2, 1.0)
    ts = TS(seg, 0, w)
    assert ts.seek(128) == 129 and ts.advance() == 130 and ts.advance() == TERMINATED
    ts = TS(seg, 0, w)
    assert ts.seek(129) == 129 and ts.advance() == 130 and ts.advance() == TERMINATED
    ts = TS(seg, 0, w)
    assert ts.doc() == 0 and ts.seek(131) == TERMINATED and ts.doc() == TERMINATED


def test_seek_against_linear():
    docs = sorted({0} | {(i * i // 100) + i for i in range(1300)})
    seg = _basic_seg(docs)
    w = O.bm25_for_one_term(1, 2, 1.0)
    arr = np.array(docs)
    for target in [0, 424, 10000, 18190, 100_000] + [3 * i for i in range(0, 6000, 37)]:
        ts = TS(seg, 0, w)
        i = int(np.searchsorted(arr, target))
        assert ts.seek(target) == (docs[i] if i < len(docs) else TERMINATED)


# ------------------------------------------------------------------ intersection (intersection.rs:344-442)
def _docs_seg(lists, max_doc=None):
    md = max_doc or (max(max(l) for l in lists if l) + 1)
    return O.build_segment(md, [[(d, 1) for d in l] for l in lists], [1] * md)


def test_intersection_kat():
    seg = _docs_seg([[1, 3, 9], [3, 4, 9, 18], [1, 5, 9, 111]])
    d, _ = O.match_all(seg, [0, 1], O.MODE_AND)
    assert d.tolist() == [3, 9]
    d, _ = O.match_all(seg, [0, 1, 2], O.MODE_AND)
    assert d.tolist() == [9]


def test_intersection_and_union_random_vs_sets():
    rng = np.random.default_rng(17)
    for _ in range(30):
        max_doc = int(rng.integers(300, 3000))
        nl = int(rng.integers(2, 5))
        lists = [np.sort(rng.choice(max_doc, size=int(rng.integers(1, max_doc)), replace=False)).tolist()
                 for _ in range(nl)]
        seg = _docs_seg(lists, max_doc)
        d, _ = O.match_all(seg, list(range(nl)), O.MODE_AND)
        assert d.tolist() == sorted(set.intersection(*[set(l) for l in lists]))
        d, _ = O.match_all(seg, list(range(nl)), O.MODE_OR)
        assert d.tolist() == sorted(set.union(*[set(l) for l in lists]))


# ------------------------------------------------------------------ end-to-end scores
def test_topdocs_droopy_tax():
    # collector/top_score_collector.rs:838-917
    seg, v = corpus_segment(["Hello happy tax payer.", "Droopy says hello happy tax payer",
                             "I like Droopy"])
    terms = [v["droopy"], v["tax"]]
    for pruned in (True, False):
        hits = O.search(seg, terms, O.MODE_OR, 4, pruned=pruned)
        assert [d for _, d in hits] == [1, 2, 0]
        for (s, _), want in zip(hits, (0.81221175, 0.5376842, 0.48527452)):
            assert np.float32(s) == np.float32(want)
        hits2 = O.search(seg, terms, O.MODE_OR, 2, pruned=pruned)
        assert [d for _, d in hits2] == [1, 2]
    merged = O.merge_top_k([(s, 0, d) for s, d in O.search(seg, terms, O.MODE_OR, 4)], 2, 4)
    assert [(o, d) for _, o, d in merged] == [(0, 0)]
    merged = O.merge_top_k([(s, 0, d) for s, d in O.search(seg, terms, O.MODE_OR, 3)], 1, 2)
    assert [(o, d) for _, o, d in merged] == [(0, 2), (0, 0)]


def test_boolean_query_with_weight():
    # boolean_query/mod.rs:221-258
    seg, v = corpus_segment(["a b c", "a c", "b c"])
    terms = [v["a"], v["b"]]
    docs, scores = O.match_all(seg, terms, O.MODE_OR)
    assert docs[0] == 0 and np.float32(scores[0]) == np.float32(0.84163445)
    ws = O.default_weights(seg, terms, O.MODE_OR)
    for w in ws:
        L.to_bm25_boost_by(C.byref(w), C.c_float(2.0))
    docs, scores = O.match_all(seg, terms, O.MODE_OR, weights=ws)
    assert nearly_equals(float(scores[0]), 1.6832689)


def test_intersection_score():
    # boolean_query/mod.rs:262-292
    seg, v = corpus_segment(["a b c", "a c", "b c", "a b c d", "d"])
    docs, scores = O.match_all(seg, [v["a"], v["b"]], O.MODE_AND)
    assert docs.tolist() == [0, 3]
    assert nearly_equals(float(scores[0]), 0.977973)
    assert nearly_equals(float(scores[1]), 0.84699446)
    hits = O.search(seg, [v["a"], v["b"]], O.MODE_AND, 10, pruned=True)
    assert [d for _, d in hits] == [0, 3]


def test_phrase_score_and_counts():
    # phrase_query/mod.rs:163-167 (asserted to 5e-4 by the reference)
    seg, v = corpus_segment(["a b c", "a b c a b"])
    docs, scores = O.match_all(seg, [v["a"], v["b"]], O.MODE_PHRASE)
    assert docs.tolist() == [0, 1]
    assert nearly_equals(float(scores[0]), 0.40618482)
    assert nearly_equals(float(scores[1]), 0.46844664)
    # phrase_weight.rs:113-133: counts 2 and 1
    seg, v = corpus_segment(["a c", "a a b d a b c", " a b"])
    docs, scores = O.match_all(seg, [v["a"], v["b"]], O.MODE_PHRASE)
    assert docs.tolist() == [1, 2]
    w = O.default_weights(seg, [v["a"], v["b"]], O.MODE_PHRASE)[0]
    assert np.float32(scores[0]) == np.float32(O.bm25_score(w, O.fieldnorm_to_id(7), 2))
    assert np.float32(scores[1]) == np.float32(O.bm25_score(w, O.fieldnorm_to_id(2), 1))


def test_phrase_position_merges():
    # phrase_scorer.rs:612-619 through one-doc phrase queries with offsets
    # [5,7] vs [1,5,10,12] with equal adjusted positions -> 1 common ; second case 3 common
    for left, right, want in (([5, 7], [1, 5, 10, 12], 1), ([1, 5, 6, 9, 10, 12], [6, 8, 9, 12], 3)):
        # term0 at positions p, term1 at positions q+1 => phrase "t0 t1" matches where p == q
        plist = [[(0, len(left))], [(0, len(right))]]
        pos = [[left], [[q + 1 for q in right]]]
        seg = O.build_segment(1, plist, [40], record_option=O.WITH_FREQS_AND_POSITIONS,
                              positions=pos)
        docs, scores = O.match_all(seg, [0, 1], O.MODE_PHRASE)
        w = O.default_weights(seg, [0, 1], O.MODE_PHRASE)[0]
        assert docs.tolist() == [0]
        assert np.float32(scores[0]) == np.float32(O.bm25_score(w, O.fieldnorm_to_id(40), want))


# ------------------------------------------------------------------ TopNHeap / merge_top_k
class TopN(C.Structure):
    _fields_ = [("heap", C.POINTER(O.Hit)), ("len", C.c_size_t), ("top_n", C.c_size_t),
                ("has_threshold", C.c_int), ("threshold", C.c_float)]


def _heap(n):
    h = TopN()
    L.to_topn_init.argtypes = [C.POINTER(TopN), C.c_size_t]
    L.to_topn_push.argtypes = [C.POINTER(TopN), C.c_float, C.c_uint32]
    L.to_topn_into_vec.restype = C.c_size_t
    L.to_topn_into_vec.argtypes = [C.POINTER(TopN), C.POINTER(O.Hit)]
    L.to_topn_init(C.byref(h), n)
    return h


def _vec(h):
    out = (O.Hit * max(1, h.top_n))()
    n = L.to_topn_into_vec(C.byref(h), out)
    return sorted([(out[i].score, out[i].doc) for i in range(n)], key=lambda x: (-x[0], x[1]))


def test_top_n_heap_units():
    # sort_by_score.rs:168-252
    h = _heap(0)
    L.to_topn_push(C.byref(h), 1.0, 0)
    L.to_topn_push(C.byref(h), 2.0, 1)
    assert _vec(h) == []
    h = _heap(2)
    for s, d in ((1.0, 0), (3.0, 1), (2.0, 2)):
        L.to_topn_push(C.byref(h), s, d)
    assert _vec(h) == [(3.0, 1), (2.0, 2)]
    h = _heap(2)
    assert not h.has_threshold
    L.to_topn_push(C.byref(h), 1.0, 0)
    assert not h.has_threshold
    L.to_topn_push(C.byref(h), 3.0, 1)
    assert h.has_threshold and h.threshold == 1.0
    L.to_topn_push(C.byref(h), 2.0, 2)
    assert h.threshold == 2.0
    L.to_topn_push(C.byref(h), 4.0, 3)
    assert h.threshold == 3.0
    h = _heap(2)
    for d in (0, 1, 2):
        L.to_topn_push(C.byref(h), 5.0, d)
    assert _vec(h) == [(5.0, 0), (5.0, 1)]
    h = _heap(1)
    L.to_topn_push(C.byref(h), 1.0, 0)
    assert h.threshold == 1.0
    L.to_topn_push(C.byref(h), 0.5, 1)
    L.to_topn_push(C.byref(h), 2.0, 2)
    assert h.threshold == 2.0 and _vec(h) == [(2.0, 2)]
    h = _heap(5)
    for s, d in ((3.0, 0), (1.0, 1), (2.0, 2)):
        L.to_topn_push(C.byref(h), s, d)
    assert not h.has_threshold and _vec(h) == [(3.0, 0), (2.0, 2), (1.0, 1)]


def test_top_n_heap_matches_full_sort():
    rng = np.random.default_rng(1)
    for _ in range(100):
        limit = int(rng.integers(0, 20))
        n = int(rng.integers(0, 200))
        docs = np.sort(rng.choice(1000, size=n, replace=False))
        scores = rng.integers(0, 50, size=n).astype(np.float32)
        h = _heap(limit)
        for s, d in zip(scores, docs):
            L.to_topn_push(C.byref(h), float(s), int(d))
        want = sorted(zip(scores.tolist(), docs.tolist()), key=lambda x: (-x[0], x[1]))[:limit]
        assert _vec(h) == want


def test_merge_top_k():
    # sort_key_top_collector.rs:164-191 (Desc order = score order)
    vals = [(float(v), 0, v) for v in range(10)]
    np.random.default_rng(0).shuffle(vals)
    assert O.merge_top_k(vals, 0, 0) == []
    assert [(s, d) for s, _, d in O.merge_top_k(vals, 0, 2)] == [(9.0, 9), (8.0, 8)]
    assert [(s, d) for s, _, d in O.merge_top_k(vals, 2, 2)] == [(7.0, 7), (6.0, 6)]
    assert len(O.merge_top_k(vals, 0, 11)) == 10
    # ties: (segment_ord, doc) ascending
    tie = [(1.0, 1, 5), (1.0, 0, 9), (1.0, 0, 3), (2.0, 3, 0)]
    assert O.merge_top_k(tie, 0, 4) == [(2.0, 3, 0), (1.0, 0, 3), (1.0, 0, 9), (1.0, 1, 5)]


# ------------------------------------------------------------------ block-WAND differential (reference regression inputs)
def _expanded(posting_lists, fieldnorms, repeat=64):
    fn = [f for f in fieldnorms for _ in range(repeat)]
    pls = [[(d * repeat + o, tf if o == 0 else 1) for d, tf in pl for o in range(repeat)]
           for pl in posting_lists]
    return pls, fn


def _bw_nearly(a, b):
    return abs(a - b) < 0.0001 * abs(a + b)


def _compare_pruned_vs_exhaustive(seg, terms, mode, ks=(1, 2, 3)):
    for k in ks:
        a = O.search(seg, terms, mode, k, pruned=True)
        b = O.search(seg, terms, mode, k, pruned=False)
        assert len(a) == len(b)
        kth = b[-1][0] if b else 0.0
        for (sa, da), (sb, db) in zip(a, b):
            assert _bw_nearly(sa, sb)
            if not _bw_nearly(sb, kth):  # the reference excludes near-threshold docs too
                assert da == db


GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "block_wand_regressions.json")))


def test_block_wand_union_regression():
    g = GOLD["union_reproduce_proptest"]
    pls, fn = _expanded([[tuple(p) for p in pl] for pl in g["posting_lists"]], g["fieldnorms"])
    seg = O.build_segment(len(fn), pls, fn)
    _compare_pruned_vs_exhaustive(seg, [0, 1, 2], O.MODE_OR)
    for single in range(3):
        _compare_pruned_vs_exhaustive(seg, [single], O.MODE_OR)


def test_block_wand_intersection_regression():
    g = GOLD["intersection_three_scorers_regression"]
    pls, fn = _expanded([[tuple(p) for p in pl] for pl in g["posting_lists"]], g["fieldnorms"])
    seg = O.build_segment(len(fn), pls, fn)
    _compare_pruned_vs_exhaustive(seg, [0, 1, 2], O.MODE_AND)
    _compare_pruned_vs_exhaustive(seg, [0, 1], O.MODE_AND)


def test_block_wand_intersection_disjoint_and_overlap():
    seg = O.build_segment(200, [[(d, 1) for d in range(100)], [(d, 1) for d in range(100, 200)]],
                          [10] * 200)
    assert O.search(seg, [0, 1], O.MODE_AND, 10) == []
    seg = O.build_segment(50, [[(d, 3) for d in range(50)]] * 2, [10] * 50)
    a = O.search(seg, [0, 1], O.MODE_AND, 5)
    b = O.search(seg, [0, 1], O.MODE_AND, 5, pruned=False)
    assert a == b and [d for _, d in a] == [0, 1, 2, 3, 4]


@pytest.mark.parametrize("nterms", [1, 2, 3])
def test_block_wand_random_differential(nterms):
    # proptest analogue of block_wand_union.rs:331-503 / block_wand_intersection.rs:218-424
    rng = np.random.default_rng(100 + nterms)
    for _ in range(40):
        max_doc = int(rng.integers(1, 100))
        fieldnorms = rng.integers(2, 1000, size=max_doc).tolist()
        pls = [random_postings(rng, max_doc, int(rng.integers(1, max_doc + 1)), 99)
               for _ in range(nterms)]
        epl, efn = _expanded(pls, fieldnorms)
        seg = O.build_segment(len(efn), epl, efn)
        _compare_pruned_vs_exhaustive(seg, list(range(nterms)), O.MODE_OR)
        if nterms >= 2:
            _compare_pruned_vs_exhaustive(seg, list(range(nterms)), O.MODE_AND)


# ------------------------------------------------------------------ synthetic generator sanity
def test_synth_segment_shapes():
    seg = O.synth_segment(20_000, n_terms=64, with_positions=True)
    assert seg.terms[0].doc_freq > seg.terms[1].doc_freq > seg.terms[63].doc_freq > 0
    assert abs(seg.terms[0].doc_freq - 10_000) < 500
    docs, tfs = O.decode_postings(seg, 0)
    assert len(docs) == seg.terms[0].doc_freq and np.all(np.diff(docs.astype(np.int64)) > 0)
    assert tfs.min() >= 1 and tfs.max() <= 10
    total = int(tfs.sum())
    flat, n = O.decode_positions(seg, 0, total)
    assert n == total
    # phrase plants make "1 2 3" match somewhere
    d, s = O.match_all(seg, [0, 1, 2], O.MODE_PHRASE)
    assert len(d) > 0
    seg2 = O.synth_segment(20_000, n_terms=64, with_positions=True)
    assert np.array_equal(seg.idx, seg2.idx) and np.array_equal(seg.pos, seg2.pos)


def test_bool_match_all_consistency():
    """The boolean restatement against the C oracle's own AND / OR paths and by brute force."""
    seg = O.synth_segment(30_000, n_terms=20, with_positions=False)
    d, sc = O.bool_match_all(seg, [1, 2, 3], [O.MUST] * 3)
    d2, sc2 = O.match_all(seg, [1, 2, 3], O.MODE_AND)
    order = np.argsort(d2)
    assert np.array_equal(d, d2[order]) and np.array_equal(sc, sc2[order])
    d, sc = O.bool_match_all(seg, [4, 5], [O.SHOULD] * 2)
    d2, sc2 = O.match_all(seg, [4, 5], O.MODE_OR)
    order = np.argsort(d2)
    assert np.array_equal(d, d2[order]) and np.array_equal(sc, sc2[order])
    # Must + Should + MustNot by sets
    d, sc = O.bool_match_all(seg, [1, 6, 7], [O.MUST, O.SHOULD, O.MUST_NOT])
    d1, s1 = O.match_all(seg, [1], O.MODE_OR)
    d6, s6 = O.match_all(seg, [6], O.MODE_OR)
    d7, _ = O.match_all(seg, [7], O.MODE_OR)
    keep = sorted(set(d1.tolist()) - set(d7.tolist()))
    assert d.tolist() == keep
    m1 = dict(zip(d1.tolist(), s1.tolist()))
    m6 = dict(zip(d6.tolist(), s6.tolist()))
    for doc, x in zip(d.tolist(), sc.tolist()):
        want = np.float32(m1[doc]) + np.float32(m6[doc]) if doc in m6 else np.float32(m1[doc])
        assert np.float32(x) == np.float32(want)
    assert len(O.bool_match_all(seg, [3], [O.MUST_NOT])[0]) == 0


def test_scorer_tree_restatement_against_the_semantic_one():
    """The C restatement of the generic scorer tree (Intersection / BufferedUnionScorer /
    Disjunction / RequiredOptionalScorer / Exclude, what the bench's cpu_baseline times for
    boolean queries) and the dense numpy restatement of the same semantics agree."""
    seg = O.synth_segment(40_000, n_terms=30, with_positions=False)
    rng = np.random.default_rng(3)
    M, S, N = O.MUST, O.SHOULD, O.MUST_NOT
    shapes = [([M, S], None, 0), ([M, N], None, 0), ([S, S, N], None, 0), ([M, M, M, N], None, 0),
              ([M, S, N, S, M], None, 0), ([N], None, 0), ([M, M, M], [0, 1, 1], 0),
              ([M, M, M, M], [0, 0, 1, 1], 0), ([M, S, S], None, 1), ([M, S, S, S], None, 2),
              ([S, S, S], None, 2), ([S, S, S], None, 3), ([M, S], None, 2),
              ([M, S, S, N], [0, 1, 1, 2], 1), ([M, M, M, M, M], [0, 1, 1, 2, 2], 0)]
    for occ, cof, msm in shapes * 2:
        terms = rng.choice(30, size=len(occ), replace=False).tolist()
        d1, s1 = O.bool_match_all(seg, terms, occ, cof, msm)
        d2, s2 = O.bool_match_all_c(seg, terms, occ, cof, msm)
        assert np.array_equal(d1, d2), (occ, cof, msm, terms)
        assert np.allclose(s1, s2, rtol=1e-5, atol=0), (occ, cof, msm, terms)
        top = O.bool_search(seg, terms, occ, 10, cof, msm)
        want = sorted(zip(s2.tolist(), d2.tolist()), key=lambda h: (-h[0], h[1]))[:10]
        assert [d for _, d in top] == [d for _, d in want]


def test_scorer_tree_reference_unit_cases():
    """exclude.rs:104-117 (test_exclude), disjunction.rs tests (docs matched by >= pass_line of the
    lists), reqopt_scorer.rs tests (the required side decides the doc set)."""
    a = [1, 2, 5, 8, 10, 15, 24]
    b = [1, 2, 3, 10, 16, 24]
    c = [2, 5, 9, 10, 24, 30]
    seg = O.build_segment(64, [[(d, 1) for d in l] for l in (a, b, c)], [3] * 64)
    M, S, N = O.MUST, O.SHOULD, O.MUST_NOT
    d, _ = O.bool_match_all_c(seg, [0, 1], [M, N])
    assert d.tolist() == [5, 8, 15]
    d, _ = O.bool_match_all_c(seg, [0, 1, 2], [S, S, S], None, 2)
    assert d.tolist() == sorted(x for x in set(a + b + c) if (x in a) + (x in b) + (x in c) >= 2)
    d, sc = O.bool_match_all_c(seg, [0, 1], [M, S])
    assert d.tolist() == a
    one = sc[a.index(5)]                       # only the required term matches doc 5
    assert all((s > one) == (doc in b) for doc, s in zip(a, sc.tolist()))


# ------------------------------------------------------------------ SSE2 decode of the CPU baseline
def test_simd_decode_equals_scalar_decode():
    """oracle/to_simd.c (the BitPacker4x decode the CPU baseline times, written with SSE2 registers
    as the reference's `bitpacking` crate is) against the scalar restatement: every bit width
    0..32, plain / minus-one tf blocks, strict and legacy doc deltas, with and without a seed."""
    rng = np.random.default_rng(20260923)
    try:
        for b in range(0, 33):
            for trial in range(8):
                hi = 1 << b
                vals = (rng.integers(0, hi, size=128, dtype=np.uint64).astype(np.uint32)
                        if b else np.zeros(128, np.uint32))
                if b:
                    vals[int(rng.integers(0, 128))] = hi - 1
                for minus_one in (False, True):
                    src = vals if not minus_one else (vals.astype(np.uint64) + 1).clip(1, 0xFFFFFFFF).astype(np.uint32)
                    nb, data = O.compress_block_unsorted(src, minus_one)
                    O.set_simd(False)
                    n0, a = O.uncompress_block_unsorted(data, nb, minus_one)
                    O.set_simd(True)
                    n1, c = O.uncompress_block_unsorted(data, nb, minus_one)
                    assert n0 == n1 == 16 * nb
                    assert np.array_equal(a, c) and np.array_equal(a, src), (b, minus_one)
        for trial in range(200):
            gaps = rng.integers(1, 1 << int(rng.integers(1, 20)), size=128)
            off = int(rng.integers(1, 1000)) if trial % 3 else 0
            docs = (off + np.cumsum(gaps)).astype(np.uint32)
            nb, data = O.compress_block_sorted(docs, off)
            O.set_simd(False)
            _, a = O.uncompress_block_sorted(data, off, nb, True)
            O.set_simd(True)
            _, c = O.uncompress_block_sorted(data, off, nb, True)
            assert np.array_equal(a, c) and np.array_equal(a, docs)
            # legacy (non-strict) deltas decode the same bytes as v[i] = v[i-1] + d[i]
            O.set_simd(False)
            _, a = O.uncompress_block_sorted(data, off, nb, False)
            O.set_simd(True)
            _, c = O.uncompress_block_sorted(data, off, nb, False)
            assert np.array_equal(a, c)
    finally:
        O.set_simd(False)


def test_simd_baseline_returns_the_scalar_top_k():
    """The executors give the same hits whichever decoder is switched in."""
    seg = O.synth_segment(150_000, n_terms=48, with_positions=True, phrase_terms=8)
    qs = [(O.MODE_AND, [0, 3]), (O.MODE_AND, [5, 40]), (O.MODE_OR, [1, 7, 9, 30, 44]),
          (O.MODE_PHRASE, [0, 1, 2])]
    try:
        for mode, terms in qs:
            O.set_simd(False)
            a = O.search(seg, terms, mode, 10, pruned=True)
            O.set_simd(True)
            b = O.search(seg, terms, mode, 10, pruned=True)
            assert a == b
    finally:
        O.set_simd(False)
