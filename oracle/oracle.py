"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (tantivy_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

TERMINATED = 0x7FFFFFFF
BLOCK_LEN = 128
BASIC, WITH_FREQS, WITH_FREQS_AND_POSITIONS = 0, 1, 2
MODE_AND, MODE_OR, MODE_PHRASE, MODE_BOOL = 0, 1, 2, 3


def build(force=False):
    srcs = ["to_codec.c", "to_simd.c", "to_postings.c", "to_query.c", "to_gen.c", "tantivy_oracle.h", "Makefile"]
    if not force and os.path.exists(_LIB_PATH):
        so_m = os.path.getmtime(_LIB_PATH)
        if all(os.path.getmtime(os.path.join(_HERE, s)) <= so_m for s in srcs):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class TermInfo(C.Structure):
    _fields_ = [
        ("doc_freq", C.c_uint32),
        ("postings_start", C.c_uint64),
        ("postings_end", C.c_uint64),
        ("positions_start", C.c_uint64),
        ("positions_end", C.c_uint64),
    ]


class Bm25(C.Structure):
    _fields_ = [("weight", C.c_float), ("cache", C.c_float * 256), ("average_fieldnorm", C.c_float)]


class SegmentView(C.Structure):
    _fields_ = [
        ("max_doc", C.c_uint32),
        ("record_option", C.c_int),
        ("idx", C.c_void_p),
        ("idx_len", C.c_size_t),
        ("pos", C.c_void_p),
        ("pos_len", C.c_size_t),
        ("fieldnorm", C.c_void_p),
        ("total_num_tokens", C.c_uint64),
    ]


class Query(C.Structure):
    _fields_ = [
        ("n_terms", C.c_uint32),
        ("terms", C.POINTER(TermInfo)),
        ("weights", C.POINTER(Bm25)),
        ("phrase_offsets", C.POINTER(C.c_uint32)),
        ("mode", C.c_int),
        ("k", C.c_uint32),
        ("occurs", C.POINTER(C.c_uint8)),
        ("clause_of", C.POINTER(C.c_uint8)),
        ("min_should_match", C.c_uint32),
    ]


class TreeNode(C.Structure):  # to_tree_node
    _fields_ = [("kind", C.c_uint8), ("occur", C.c_uint8), ("n_kids", C.c_uint16), ("first", C.c_uint32),
                ("msm", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("score", C.c_float), ("doc", C.c_uint32)]


class GlobalHit(C.Structure):
    _fields_ = [("score", C.c_float), ("segment_ord", C.c_uint32), ("doc", C.c_uint32)]


class SynthSegment(C.Structure):
    _fields_ = [
        ("max_doc", C.c_uint32),
        ("n_terms", C.c_uint32),
        ("record_option", C.c_int),
        ("idx", Buf),
        ("pos", Buf),
        ("fieldnorm", Buf),
        ("terms", C.POINTER(TermInfo)),
        ("total_num_tokens", C.c_uint64),
    ]


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.to_id_to_fieldnorm.restype = C.c_uint32
    L.to_id_to_fieldnorm.argtypes = [C.c_uint8]
    L.to_fieldnorm_to_id.restype = C.c_uint8
    L.to_fieldnorm_to_id.argtypes = [C.c_uint32]
    L.to_fieldnorm_table.restype = u32p
    L.to_vint_serialize.restype = C.c_size_t
    L.to_vint_serialize.argtypes = [C.c_uint64, u8p]
    L.to_vint_deserialize.restype = C.c_size_t
    L.to_vint_deserialize.argtypes = [u8p, C.c_size_t, C.POINTER(C.c_uint64)]
    for name in ("to_vint_compress_sorted",):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [u32p, C.c_size_t, u8p, C.c_uint32]
    L.to_vint_compress_unsorted.restype = C.c_size_t
    L.to_vint_compress_unsorted.argtypes = [u32p, C.c_size_t, u8p]
    L.to_vint_uncompress_sorted.restype = C.c_size_t
    L.to_vint_uncompress_sorted.argtypes = [u8p, u32p, C.c_size_t, C.c_uint32]
    L.to_vint_uncompress_unsorted.restype = C.c_size_t
    L.to_vint_uncompress_unsorted.argtypes = [u8p, u32p, C.c_size_t]
    L.to_vint_uncompress_unsorted_until_end.restype = C.c_size_t
    L.to_vint_uncompress_unsorted_until_end.argtypes = [u8p, C.c_size_t, u32p, C.c_size_t]
    L.to_compress_block_sorted.restype = C.c_uint8
    L.to_compress_block_sorted.argtypes = [u32p, C.c_uint32, u8p, C.POINTER(C.c_size_t)]
    L.to_compress_block_unsorted.restype = C.c_uint8
    L.to_compress_block_unsorted.argtypes = [u32p, C.c_int, u8p, C.POINTER(C.c_size_t)]
    L.to_uncompress_block_sorted.restype = C.c_size_t
    L.to_uncompress_block_sorted.argtypes = [u8p, C.c_uint32, C.c_uint8, C.c_int, u32p]
    L.to_uncompress_block_unsorted.restype = C.c_size_t
    L.to_uncompress_block_unsorted.argtypes = [u8p, C.c_uint8, C.c_int, u32p]
    L.to_search_block.restype = C.c_size_t
    L.to_search_block.argtypes = [u32p, C.c_uint32]
    L.to_idf.restype = C.c_float
    L.to_idf.argtypes = [C.c_uint64, C.c_uint64]
    L.to_bm25_new.argtypes = [C.POINTER(Bm25), C.c_float, C.c_float]
    L.to_bm25_for_one_term.argtypes = [C.POINTER(Bm25), C.c_uint64, C.c_uint64, C.c_float]
    L.to_bm25_boost_by.argtypes = [C.POINTER(Bm25), C.c_float]
    L.to_bm25_score.restype = C.c_float
    L.to_bm25_score.argtypes = [C.POINTER(Bm25), C.c_uint8, C.c_uint32]
    L.to_bm25_tf_factor.restype = C.c_float
    L.to_bm25_tf_factor.argtypes = [C.POINTER(Bm25), C.c_uint8, C.c_uint32]
    L.to_bm25_max_score.restype = C.c_float
    L.to_bm25_max_score.argtypes = [C.POINTER(Bm25)]
    L.to_encode_bitwidth.restype = C.c_uint8
    L.to_encode_bitwidth.argtypes = [C.c_uint8, C.c_int]
    L.to_encode_block_wand_max_tf.restype = C.c_uint8
    L.to_encode_block_wand_max_tf.argtypes = [C.c_uint32]
    L.to_decode_block_wand_max_tf.restype = C.c_uint32
    L.to_decode_block_wand_max_tf.argtypes = [C.c_uint8]
    L.to_buf_init.argtypes = [C.POINTER(Buf)]
    L.to_buf_free.argtypes = [C.POINTER(Buf)]
    L.to_postings_serializer_new.restype = C.c_void_p
    L.to_postings_serializer_new.argtypes = [C.c_float, C.c_int, C.c_void_p, C.c_uint32]
    L.to_postings_serializer_free.argtypes = [C.c_void_p]
    L.to_postings_serializer_new_term.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.to_postings_serializer_write_doc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.to_postings_serializer_close_term.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Buf)]
    L.to_position_serializer_new.restype = C.c_void_p
    L.to_position_serializer_new.argtypes = [C.POINTER(Buf)]
    L.to_position_serializer_free.argtypes = [C.c_void_p]
    L.to_position_serializer_write_positions_delta.argtypes = [C.c_void_p, u32p, C.c_size_t]
    L.to_position_serializer_close_term.argtypes = [C.c_void_p]
    L.to_serialize_postings_batch.argtypes = [C.c_float, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Buf),
                                              C.c_void_p]
    L.to_serialize_positions_batch.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Buf),
                                               C.c_void_p]
    L.to_search_pruned.restype = C.c_size_t
    L.to_search_pruned.argtypes = [C.POINTER(SegmentView), C.POINTER(Query), C.POINTER(Hit)]
    L.to_search_exhaustive.restype = C.c_size_t
    L.to_search_exhaustive.argtypes = [C.POINTER(SegmentView), C.POINTER(Query), C.POINTER(Hit)]
    L.to_match_all.restype = C.c_size_t
    L.to_match_all.argtypes = [C.POINTER(SegmentView), C.POINTER(Query), u32p, f32p, C.c_size_t]
    L.to_tree_match_all.restype = C.c_size_t
    L.to_tree_match_all.argtypes = [C.POINTER(SegmentView), C.POINTER(TreeNode), C.c_size_t, C.POINTER(TermInfo),
                                    C.POINTER(Bm25), u32p, C.c_size_t, u32p, f32p, C.c_size_t]
    L.to_sort_hits.argtypes = [C.POINTER(Hit), C.c_size_t]
    L.to_decode_postings.restype = C.c_size_t
    L.to_decode_postings.argtypes = [C.POINTER(SegmentView), C.POINTER(TermInfo), u32p, u32p]
    L.to_decode_positions.restype = C.c_size_t
    L.to_decode_positions.argtypes = [C.POINTER(SegmentView), C.POINTER(TermInfo), u32p, C.c_size_t]
    L.to_merge_top_k.restype = C.c_size_t
    L.to_merge_top_k.argtypes = [C.POINTER(GlobalHit), C.c_size_t, C.c_size_t, C.c_size_t,
                                 C.POINTER(GlobalHit)]
    L.to_synth_build.restype = C.POINTER(SynthSegment)
    L.to_synth_build.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.to_synth_free.argtypes = [C.POINTER(SynthSegment)]
    L.to_rng_seed.argtypes = [C.POINTER(Rng), C.c_uint64]
    L.to_rng_next.restype = C.c_uint64
    L.to_rng_next.argtypes = [C.POINTER(Rng)]
    L.to_rng_uniform.restype = C.c_double
    L.to_rng_uniform.argtypes = [C.POINTER(Rng)]
    L.to_zipf_rank.restype = C.c_uint32
    L.to_zipf_rank.argtypes = [C.POINTER(Rng), C.c_uint32]
    L.to_baseline_run.restype = C.c_double
    L.to_baseline_run.argtypes = [C.POINTER(SegmentView), C.POINTER(Query), C.c_size_t, C.c_int,
                                  C.POINTER(C.c_double), C.POINTER(Hit), u32p]
    _lib = L
    return L


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


# ----------------------------------------------------------------------------- codec helpers
def set_simd(on):
    """Baseline runs: decode posting blocks with the SSE2 unpack of to_simd.c (the reference's
    BitPacker4x is SIMD code) instead of the scalar loops.  Returns the previous setting."""
    L = lib()
    prev = bool(L.to_get_simd())
    L.to_set_simd(1 if on else 0)
    return prev


def compress_block_sorted(vals, offset):
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    assert v.size == BLOCK_LEN
    out = np.zeros(BLOCK_LEN * 5, dtype=np.uint8)
    n = C.c_size_t()
    nb = lib().to_compress_block_sorted(_u32(v), int(offset), _u8(out), C.byref(n))
    return nb, out[: n.value].copy()


def compress_block_unsorted(vals, minus_one):
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    assert v.size == BLOCK_LEN
    out = np.zeros(BLOCK_LEN * 5, dtype=np.uint8)
    n = C.c_size_t()
    nb = lib().to_compress_block_unsorted(_u32(v), int(bool(minus_one)), _u8(out), C.byref(n))
    return nb, out[: n.value].copy()


def uncompress_block_sorted(data, offset, num_bits, strict=True):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    pad = np.concatenate([d, np.zeros(32, np.uint8)])
    out = np.zeros(BLOCK_LEN, dtype=np.uint32)
    n = lib().to_uncompress_block_sorted(_u8(pad), int(offset), int(num_bits), int(strict), _u32(out))
    return n, out


def uncompress_block_unsorted(data, num_bits, minus_one):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    pad = np.concatenate([d, np.zeros(32, np.uint8)])
    out = np.zeros(BLOCK_LEN, dtype=np.uint32)
    n = lib().to_uncompress_block_unsorted(_u8(pad), int(num_bits), int(bool(minus_one)), _u32(out))
    return n, out


def vint_compress_sorted(vals, offset):
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    out = np.zeros(max(1, v.size) * 5, dtype=np.uint8)
    n = lib().to_vint_compress_sorted(_u32(v), v.size, _u8(out), int(offset))
    return out[:n].copy()


def vint_compress_unsorted(vals):
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    out = np.zeros(max(1, v.size) * 5, dtype=np.uint8)
    n = lib().to_vint_compress_unsorted(_u32(v), v.size, _u8(out))
    return out[:n].copy()


def vint_uncompress_sorted(data, n, offset):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint32)
    c = lib().to_vint_uncompress_sorted(_u8(d), _u32(out), n, int(offset))
    return c, out


def vint_uncompress_unsorted(data, n):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint32)
    c = lib().to_vint_uncompress_unsorted(_u8(d), _u32(out), n)
    return c, out


def fieldnorm_table():
    p = lib().to_fieldnorm_table()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()


def fieldnorm_to_id(v):
    return lib().to_fieldnorm_to_id(int(v) & 0xFFFFFFFF)


# ----------------------------------------------------------------------------- BM25
def bm25_for_one_term(term_doc_freq, total_num_docs, avg_fieldnorm, boost=1.0):
    w = Bm25()
    lib().to_bm25_for_one_term(C.byref(w), int(term_doc_freq), int(total_num_docs),
                               C.c_float(avg_fieldnorm))
    if boost != 1.0:
        lib().to_bm25_boost_by(C.byref(w), C.c_float(boost))
    return w


def bm25_for_terms(term_doc_freqs, total_num_docs, avg_fieldnorm):
    """Bm25Weight::for_terms with several terms: idf summed (bm25.rs:121-128)."""
    idf_sum = np.float32(0.0)
    for df in term_doc_freqs:
        idf_sum = np.float32(idf_sum + np.float32(lib().to_idf(int(df), int(total_num_docs))))
    w = Bm25()
    lib().to_bm25_new(C.byref(w), C.c_float(float(idf_sum)), C.c_float(avg_fieldnorm))
    return w


def bm25_score(w, fieldnorm_id, tf):
    return lib().to_bm25_score(C.byref(w), int(fieldnorm_id), int(tf) & 0xFFFFFFFF)


def bm25_max_score(w):
    return lib().to_bm25_max_score(C.byref(w))


# ----------------------------------------------------------------------------- segments
class Segment:
    """A single-field segment in tantivy's byte format, held in numpy arrays.

    idx: field .idx sub-file (8-byte total_num_tokens header + posting lists)
    pos: field .pos sub-file (may be empty);  fieldnorm: max_doc bytes (or None)
    terms: list of TermInfo
    """

    def __init__(self, max_doc, record_option, idx, pos, fieldnorm, terms, total_num_tokens):
        self.max_doc = int(max_doc)
        self.record_option = int(record_option)
        # pad so that block decoders may over-read a few bytes
        self.idx = np.concatenate([np.ascontiguousarray(idx, np.uint8), np.zeros(64, np.uint8)])
        self.idx_len = int(len(idx))
        self.pos = np.concatenate([np.ascontiguousarray(pos, np.uint8), np.zeros(64, np.uint8)])
        self.pos_len = int(len(pos))
        self.fieldnorm = None if fieldnorm is None else np.ascontiguousarray(fieldnorm, np.uint8)
        self.terms = terms
        self.total_num_tokens = int(total_num_tokens)
        v = SegmentView()
        v.max_doc = self.max_doc
        v.record_option = self.record_option
        v.idx = self.idx.ctypes.data
        v.idx_len = self.idx_len
        v.pos = self.pos.ctypes.data if self.pos_len else None
        v.pos_len = self.pos_len
        v.fieldnorm = None if self.fieldnorm is None else self.fieldnorm.ctypes.data
        v.total_num_tokens = self.total_num_tokens
        self.view = v

    @property
    def avg_fieldnorm(self):
        return float(np.float32(self.total_num_tokens) / np.float32(self.max_doc))

    def term_info(self, i):
        return self.terms[i]

    def postings_bytes(self, i):
        t = self.terms[i]
        return self.idx[8 + t.postings_start: 8 + t.postings_end]


def synth_segment(max_doc, n_terms=256, segment_ord=0, with_positions=False, phrase_terms=32):
    L = lib()
    p = L.to_synth_build(int(max_doc), int(n_terms), int(segment_ord), int(bool(with_positions)),
                         int(phrase_terms))
    s = p.contents
    idx = np.ctypeslib.as_array(s.idx.data, shape=(s.idx.len,)).copy()
    pos = (np.ctypeslib.as_array(s.pos.data, shape=(s.pos.len,)).copy()
           if s.pos.len else np.zeros(0, np.uint8))
    fn = np.ctypeslib.as_array(s.fieldnorm.data, shape=(s.fieldnorm.len,)).copy()
    terms = []
    for i in range(s.n_terms):
        t = s.terms[i]
        ti = TermInfo(t.doc_freq, t.postings_start, t.postings_end, t.positions_start,
                      t.positions_end)
        terms.append(ti)
    seg = Segment(s.max_doc, s.record_option, idx, pos, fn, terms, s.total_num_tokens)
    L.to_synth_free(p)
    return seg


def build_segment(max_doc, postings, fieldnorms=None, record_option=WITH_FREQS, positions=None,
                  total_num_tokens=None, avg_fieldnorm=None):
    """Serialize hand-made posting lists through the oracle's PostingsSerializer.

    postings: list (one per term) of [(doc, tf), ...] sorted by doc.
    fieldnorms: list of u32 fieldnorm values per doc (quantised like the reference) or None.
    positions: optional list (per term) of list (per doc) of absolute sorted positions.
    """
    L = lib()
    if fieldnorms is not None:
        fn_ids = np.array([fieldnorm_to_id(f) for f in fieldnorms], dtype=np.uint8)
        if total_num_tokens is None:
            total_num_tokens = int(sum(int(f) for f in fieldnorms))
    else:
        fn_ids = None
        if total_num_tokens is None:
            total_num_tokens = 0
    if avg_fieldnorm is None:
        avg_fieldnorm = (float(np.float32(total_num_tokens) / np.float32(max_doc))
                         if (fn_ids is not None and max_doc) else 0.0)
    out = Buf()
    L.to_buf_init(C.byref(out))
    posbuf = Buf()
    L.to_buf_init(C.byref(posbuf))
    ser = L.to_postings_serializer_new(C.c_float(avg_fieldnorm), record_option,
                                       None if fn_ids is None else fn_ids.ctypes.data, max_doc)
    pser = L.to_position_serializer_new(C.byref(posbuf)) if positions is not None else None
    hdr = np.frombuffer(int(total_num_tokens).to_bytes(8, "little"), dtype=np.uint8)
    terms = []
    for ti, plist in enumerate(postings):
        start = out.len
        pstart = posbuf.len
        L.to_postings_serializer_new_term(ser, len(plist), 1)
        for di, (doc, tf) in enumerate(plist):
            if pser is not None:
                ps = positions[ti][di]
                assert len(ps) == tf
                deltas = np.diff(np.array([0] + list(ps), dtype=np.int64)).astype(np.uint32)
                deltas = np.ascontiguousarray(deltas)
                L.to_position_serializer_write_positions_delta(pser, _u32(deltas), len(deltas))
            L.to_postings_serializer_write_doc(ser, int(doc), int(tf))
        L.to_postings_serializer_close_term(ser, len(plist), C.byref(out))
        if pser is not None:
            L.to_position_serializer_close_term(pser)
        terms.append(TermInfo(len(plist), start, out.len, pstart, posbuf.len))
    body = np.ctypeslib.as_array(out.data, shape=(out.len,)).copy() if out.len else np.zeros(0, np.uint8)
    pos = (np.ctypeslib.as_array(posbuf.data, shape=(posbuf.len,)).copy()
           if posbuf.len else np.zeros(0, np.uint8))
    L.to_postings_serializer_free(ser)
    if pser is not None:
        L.to_position_serializer_free(pser)
    L.to_buf_free(C.byref(out))
    L.to_buf_free(C.byref(posbuf))
    idx = np.concatenate([hdr, body])
    return Segment(max_doc, record_option, idx, pos, fn_ids, terms, total_num_tokens)


def serialize_postings_batch(term_starts, docs, tfs, fieldnorm_ids, num_docs, avg_fieldnorm,
                             record_option):
    """PostingsSerializer over flat arrays (term t = docs/tfs[term_starts[t]:term_starts[t+1]]):
    (bytes, u64 term starts in them)."""
    ts = np.ascontiguousarray(term_starts, np.uint64)
    docs = np.ascontiguousarray(docs, np.uint32)
    tfs = None if tfs is None else np.ascontiguousarray(tfs, np.uint32)
    fn = None if fieldnorm_ids is None else np.ascontiguousarray(fieldnorm_ids, np.uint8)
    out = Buf()
    lib().to_buf_init(C.byref(out))
    ots = np.zeros(len(ts), np.uint64)
    lib().to_serialize_postings_batch(C.c_float(avg_fieldnorm), record_option,
                                      None if fn is None else fn.ctypes.data, num_docs, len(ts) - 1,
                                      ts.ctypes.data, docs.ctypes.data,
                                      None if tfs is None else tfs.ctypes.data, C.byref(out),
                                      ots.ctypes.data)
    body = np.ctypeslib.as_array(out.data, shape=(out.len,)).copy() if out.len else np.zeros(0, np.uint8)
    lib().to_buf_free(C.byref(out))
    return body, ots


def serialize_positions_batch(term_starts, deltas):
    """PositionSerializer over flat arrays of within-document position deltas."""
    ts = np.ascontiguousarray(term_starts, np.uint64)
    deltas = np.ascontiguousarray(deltas, np.uint32)
    out = Buf()
    lib().to_buf_init(C.byref(out))
    ots = np.zeros(len(ts), np.uint64)
    lib().to_serialize_positions_batch(len(ts) - 1, ts.ctypes.data, deltas.ctypes.data,
                                       C.byref(out), ots.ctypes.data)
    body = np.ctypeslib.as_array(out.data, shape=(out.len,)).copy() if out.len else np.zeros(0, np.uint8)
    lib().to_buf_free(C.byref(out))
    return body, ots


# ----------------------------------------------------------------------------- files & term dictionary
# Pure-Python restatements (small inputs only) of the file framing around the hot path and of the
# TermInfoStore, pinned by the reference's compat fixture (tests/golden/compat_index.json).
def _vint(data, at):
    """common/src/vint.rs:180-196: 7-bit groups, stop bit 0x80 on the LAST byte."""
    r, shift = 0, 0
    while True:
        b = data[at]
        at += 1
        r |= (b & 127) << shift
        if b & 128:
            return r, at
        shift += 7


def strip_footer(data):
    """src/directory/footer.rs:17-47: body | json | u32 json_len | u32 magic 1337."""
    data = bytes(data)
    assert int.from_bytes(data[-4:], "little") == 1337
    n = int.from_bytes(data[-8:-4], "little")
    return data[: len(data) - 8 - n], data[len(data) - 8 - n: len(data) - 8]


def composite_fields(body):
    """src/directory/composite_file.rs:68-84,109-150: {(field, idx): bytes}."""
    body = bytes(body)
    flen = int.from_bytes(body[-4:], "little")
    start = len(body) - 4 - flen
    n, at = _vint(body, start)
    addrs, offs, off = [], [], 0
    for _ in range(n):
        d, at = _vint(body, at)
        off += d
        field = int.from_bytes(body[at:at + 4], "little")
        idx, at = _vint(body, at + 4)
        addrs.append((field, idx))
        offs.append(off)
    offs.append(start)
    return {a: body[offs[i]:offs[i + 1]] for i, a in enumerate(addrs)}


def term_dictionary_parts(field_bytes):
    """TermDictionary::open: the wrapper strips a u32 dictionary type (1 = Fst,
    src/termdict/mod.rs:82-98), the fst dictionary is fst | term info store | u64 store_len |
    u32 fst version (src/termdict/fst_termdict/termdict.rs:123-140).
    Returns (fst bytes, store bytes)."""
    b = bytes(field_bytes)
    assert int.from_bytes(b[-4:], "little") == 1, "not an fst term dictionary"
    assert int.from_bytes(b[-8:-4], "little") == 1, "fst dictionary version"
    store_len = int.from_bytes(b[-16:-8], "little")
    main = b[:-16]
    return main[: len(main) - store_len], main[len(main) - store_len:]


def compute_num_bits(n):
    """bitpacker/src/lib.rs: bit length, but 64 once it exceeds 56."""
    a = int(n).bit_length()
    return a if a <= 56 else 64


TERM_INFO_BLOCK_LEN = 256
_BLOCK_META_SIZE = 8 + 28 + 3  # term_info_store.rs:51-53, term_info.rs:37


def _extract_bits(data, addr_bits, num_bits):
    """term_info_store.rs:108-128"""
    assert num_bits <= 56
    a = addr_bits // 8
    v = int.from_bytes(data[a:a + 8].ljust(8, b"\0"), "little")
    return (v >> (addr_bits % 8)) & ((1 << num_bits) - 1)


def term_info_store_num_terms(store):
    return int.from_bytes(bytes(store[8:16]), "little")


def term_info_store_get(store, term_ord):
    """TermInfoStore::open + get (term_info_store.rs:130-163, :64-101):
    (doc_freq, postings_start, postings_end, positions_start, positions_end)."""
    store = bytes(store)
    meta_len = int.from_bytes(store[0:8], "little")
    metas, infos = store[16:16 + meta_len], store[16 + meta_len:]
    m = metas[(term_ord // TERM_INFO_BLOCK_LEN) * _BLOCK_META_SIZE:][:_BLOCK_META_SIZE]
    offset = int.from_bytes(m[0:8], "little")
    df = int.from_bytes(m[8:12], "little")
    ps = int.from_bytes(m[12:20], "little")
    pn = int.from_bytes(m[20:24], "little")
    qs = int.from_bytes(m[24:32], "little")
    qn = int.from_bytes(m[32:36], "little")
    df_bits, post_bits, pos_bits = m[36], m[37], m[38]
    inner = term_ord % TERM_INFO_BLOCK_LEN
    if inner == 0:
        return (df, ps, ps + pn, qs, qs + qn)
    data = infos[offset:]
    nb = df_bits + post_bits + pos_bits
    a = nb * (inner - 1)
    p0 = ps + _extract_bits(data, a, post_bits)
    p1 = ps + _extract_bits(data, a + nb, post_bits)
    q0 = qs + _extract_bits(data, a + post_bits, pos_bits)
    q1 = qs + _extract_bits(data, a + post_bits + nb, pos_bits)
    d = _extract_bits(data, a + post_bits + pos_bits, df_bits)
    return (d, p0, p1, q0, q1)


def term_info_store_serialize(term_infos):
    """TermInfoStoreWriter (term_info_store.rs:171-290) over (doc_freq, postings_start,
    postings_end, positions_start, positions_end) tuples in term-ordinal order."""
    metas, infos = bytearray(), bytearray()

    class _BP:  # bitpacker/src/bitpacker.rs:24-58
        def __init__(self):
            self.buf, self.n = 0, 0

        def write(self, val, bits, out):
            self.buf |= (val & ((1 << 64) - 1)) << self.n
            self.n += bits
            while self.n >= 64:
                out += (self.buf & ((1 << 64) - 1)).to_bytes(8, "little")
                self.buf >>= 64
                self.n -= 64

        def flush(self, out):
            if self.n:
                out += (self.buf & ((1 << 64) - 1)).to_bytes(8, "little")[: (self.n + 7) // 8]
            self.buf, self.n = 0, 0

    for b0 in range(0, len(term_infos), TERM_INFO_BLOCK_LEN):
        blk = term_infos[b0:b0 + TERM_INFO_BLOCK_LEN]
        ref, last = blk[0], blk[-1]
        post_end = last[2] - ref[1]
        pos_end = last[4] - ref[3]
        df_bits = compute_num_bits(max([t[0] for t in blk[1:]], default=0))
        post_bits, pos_bits = compute_num_bits(post_end), compute_num_bits(pos_end)
        metas += len(infos).to_bytes(8, "little")
        metas += ref[0].to_bytes(4, "little") + ref[1].to_bytes(8, "little")
        metas += (ref[2] - ref[1]).to_bytes(4, "little") + ref[3].to_bytes(8, "little")
        metas += (ref[4] - ref[3]).to_bytes(4, "little") + bytes([df_bits, post_bits, pos_bits])
        bp = _BP()
        for t in blk[1:]:
            bp.write(t[1] - ref[1], post_bits, infos)
            bp.write(t[3] - ref[3], pos_bits, infos)
            bp.write(t[0], df_bits, infos)
        bp.write(post_end, post_bits, infos)
        bp.write(pos_end, pos_bits, infos)
        bp.flush(infos)
    return (len(metas).to_bytes(8, "little") + len(term_infos).to_bytes(8, "little")
            + bytes(metas) + bytes(infos))


# ----------------------------------------------------------------------------- queries
class QuerySpec:
    """Keeps the ctypes arrays alive for one to_query."""

    def __init__(self, seg, term_ids, weights, mode, k, phrase_offsets=None, occurs=None,
                 clause_of=None, min_should_match=0):
        n = len(term_ids)
        self.terms = (TermInfo * n)(*[seg.terms[t] for t in term_ids])
        self.weights = (Bm25 * len(weights))(*weights)
        self.offsets = None
        q = Query()
        q.n_terms = n
        q.terms = C.cast(self.terms, C.POINTER(TermInfo))
        q.weights = C.cast(self.weights, C.POINTER(Bm25))
        if phrase_offsets is not None:
            self.offsets = (C.c_uint32 * n)(*phrase_offsets)
            q.phrase_offsets = C.cast(self.offsets, C.POINTER(C.c_uint32))
        q.mode = mode
        q.k = k
        if occurs is not None:  # MODE_BOOL
            self.occurs = (C.c_uint8 * n)(*[int(o) for o in occurs])
            q.occurs = C.cast(self.occurs, C.POINTER(C.c_uint8))
            if clause_of is not None:
                self.clause_of = (C.c_uint8 * n)(*[int(c) for c in clause_of])
                q.clause_of = C.cast(self.clause_of, C.POINTER(C.c_uint8))
            q.min_should_match = int(min_should_match)
        self.q = q


def default_weights(seg, term_ids, mode, total_num_docs=None, total_num_tokens=None, dfs=None,
                    boosts=None):
    """Bm25Weight per term from (possibly global) statistics, as Searcher supplies them; boosts =
    the factors handed down by BoostWeight::scorer (boost_query.rs:70-72), applied with
    Bm25Weight::boost_by."""
    nd = seg.max_doc if total_num_docs is None else total_num_docs
    nt = seg.total_num_tokens if total_num_tokens is None else total_num_tokens
    avg = float(np.float32(nt) / np.float32(nd))
    dfl = [seg.terms[t].doc_freq for t in term_ids] if dfs is None else dfs
    if mode == MODE_PHRASE:
        return [bm25_for_terms(dfl, nd, avg)]
    if boosts is None:
        boosts = [1.0] * len(dfl)
    return [bm25_for_one_term(df, nd, avg, float(np.float32(b))) for df, b in zip(dfl, boosts)]


def _hits(arr, n):
    return [(float(arr[i].score), int(arr[i].doc)) for i in range(n)]


def search(seg, term_ids, mode, k, weights=None, pruned=True, phrase_offsets=None, sort=True):
    if weights is None:
        weights = default_weights(seg, term_ids, mode)
    if mode == MODE_PHRASE and phrase_offsets is None:
        phrase_offsets = list(range(len(term_ids)))
    spec = QuerySpec(seg, term_ids, weights, mode, k, phrase_offsets)
    out = (Hit * max(1, k))()
    fn = lib().to_search_pruned if pruned else lib().to_search_exhaustive
    n = fn(C.byref(seg.view), C.byref(spec.q), out)
    if sort:
        lib().to_sort_hits(out, n)
    return _hits(out, n)


def match_all(seg, term_ids, mode, weights=None, phrase_offsets=None, cap=None):
    if weights is None:
        weights = default_weights(seg, term_ids, mode)
    if mode == MODE_PHRASE and phrase_offsets is None:
        phrase_offsets = list(range(len(term_ids)))
    spec = QuerySpec(seg, term_ids, weights, mode, 1, phrase_offsets)
    if cap is None:
        if mode == MODE_OR:
            cap = sum(seg.terms[t].doc_freq for t in term_ids)
        else:
            cap = min(seg.terms[t].doc_freq for t in term_ids)
    cap = max(1, int(cap))
    docs = np.zeros(cap, np.uint32)
    scores = np.zeros(cap, np.float32)
    n = lib().to_match_all(C.byref(seg.view), C.byref(spec.q), _u32(docs),
                           scores.ctypes.data_as(C.POINTER(C.c_float)), cap)
    n = min(n, cap)
    return docs[:n], scores[:n]


SHOULD, MUST, MUST_NOT = 0, 1, 2  # src/query/occur.rs


def bool_match_all(seg, term_ids, occurs, clause_of=None, min_should_match=0):
    """BooleanQuery whose clauses are terms or unions of terms (terms sharing a clause_of value),
    restated from BooleanWeight::complex_scorer (src/query/boolean_query/boolean_weight.rs:236-431):
      - a union clause scores the sum of its matching terms in clause order (SumCombiner; the
        reference's own order depends on scorer removal, so sums of 3+ terms compare within
        1e-5) and costs the sum of its terms' doc freqs (BufferedUnionScorer::cost,
        src/query/union/buffered_union.rs:326-328);
      - Must clauses intersect, cheapest first (intersect_scorers, src/query/intersection.rs:31);
        Intersection::score = left + right + sum(others) (:325-329);
      - minimum_number_should_match (:272-305): > number of Should clauses matches nothing;
        == all of them (>= 2) turns them into Must clauses; 1 makes the union required
        (intersected with the Must part); else scorer_disjunction;
      - Should clauses are optional when there is a Must (RequiredOptionalScorer::score =
        req + opt, src/query/reqopt_scorer.rs:85-98) and form the union otherwise;
      - MustNot clauses exclude (src/query/exclude.rs); MustNot clauses alone match nothing.
    Every term scores with its own Bm25Weight (TermQuery::specialized_weight).
    Returns (docs ascending, f32 scores)."""
    md = seg.max_doc
    per = {}
    for t in set(term_ids):
        if seg.terms[t].doc_freq == 0:
            per[t] = (np.zeros(0, np.int64), np.zeros(0, np.float32))
            continue
        d, sc = match_all(seg, [t], MODE_OR)
        per[t] = (d.astype(np.int64), sc.astype(np.float32))
    if clause_of is None:
        clause_of = list(range(len(term_ids)))
    clauses = []  # [occur, [terms]] in order of first appearance
    ids = []
    for t, o, c in zip(term_ids, occurs, clause_of):
        if c not in ids:
            ids.append(c)
            clauses.append([o, []])
        assert clauses[ids.index(c)][0] == o
        clauses[ids.index(c)][1].append(t)

    def clause_eval(terms):
        hit = np.zeros(md, bool)
        sc = np.zeros(md, np.float32)
        for t in terms:
            one = np.zeros(md, np.float32)
            one[per[t][0]] = per[t][1]
            hit[per[t][0]] = True
            sc = (sc + one).astype(np.float32)
        cost = sum(seg.terms[t].doc_freq for t in terms)
        return hit, sc, cost

    none = (np.zeros(0, np.uint32), np.zeros(0, np.float32))
    must = [clause_eval(ts) for o, ts in clauses if o == MUST]
    if any(not h.any() for h, _, _ in must):
        return none
    should = [clause_eval(ts) for o, ts in clauses if o == SHOULD]
    should = [c for c in should if c[0].any()]          # EmptyScorers are removed (:255-257)
    mustnot = [clause_eval(ts) for o, ts in clauses if o == MUST_NOT]
    msm = min_should_match
    if msm > len(should):
        return none
    if msm >= 2 and msm == len(should):
        must += should
        should, msm = [], 0
    n_hit = np.zeros(md, np.int32)
    opt = np.zeros(md, np.float32)
    for hit, sc, _ in should:
        opt = (opt + sc).astype(np.float32)
        n_hit += hit
    if must:
        must.sort(key=lambda c: c[2])   # stable, like sort_by_key(cost)
        match = np.ones(md, bool)
        for hit, _, _ in must:
            match &= hit
        score = must[0][1]
        if len(must) > 1:
            score = (score + must[1][1]).astype(np.float32)
        if len(must) > 2:
            oth = np.zeros(md, np.float32)
            for _, sc, _ in must[2:]:
                oth = (oth + sc).astype(np.float32)
            score = (score + oth).astype(np.float32)
        if should:
            score = (score + opt).astype(np.float32)
            if msm:
                match &= n_hit >= msm
    elif should:
        match, score = n_hit >= max(1, msm), opt
    else:
        return none
    for hit, _, _ in mustnot:
        match &= ~hit
    docs = np.nonzero(match)[0]
    return docs.astype(np.uint32), score[docs]


def _combine_scorers(md, items, msm):
    """BooleanWeight::complex_scorer (src/query/boolean_query/boolean_weight.rs:236-431) over already evaluated
    sub-scorers: items = [(occur, (hit bool[md], score f32[md], cost))].  Returns (match, score, cost) or None for
    an EmptyScorer.  Must scorers intersect cheapest first, Intersection::score = left + right + sum(others)
    (intersection.rs:31,325-329); Should scorers: removed when empty (:255-257), minimum_number_should_match
    (:272-305) above their number matches nothing, == all of them (>= 2) makes them Must, else a union /
    disjunction that is required (no Must, or minimum >= 1) or optional (RequiredOptionalScorer::score = req + opt,
    reqopt_scorer.rs:85-98); MustNot scorers exclude (exclude.rs); MustNot alone matches nothing (:340-349).
    cost: an intersection costs its cheapest member (size_hint of the left scorer), a union the sum of its members
    (buffered_union.rs:326-328)."""
    must = [x for o, x in items if o == MUST]
    if any(x is None or not x[0].any() for x in must):
        return None
    should = [x for o, x in items if o == SHOULD and x is not None and x[0].any()]
    mustnot = [x for o, x in items if o == MUST_NOT and x is not None]
    if msm > len(should):
        return None
    if msm >= 2 and msm == len(should):
        must, should, msm = must + should, [], 0
    n_hit = np.zeros(md, np.int32)
    opt = np.zeros(md, np.float32)
    for hit, sc, _ in should:
        opt = (opt + np.where(hit, sc, np.float32(0))).astype(np.float32)
        n_hit += hit
    if must:
        must = sorted(must, key=lambda c: c[2])  # stable, like sort_by_key(cost)
        match = np.ones(md, bool)
        for hit, _, _ in must:
            match &= hit
        score = must[0][1]
        if len(must) > 1:
            score = (score + must[1][1]).astype(np.float32)
        if len(must) > 2:
            oth = np.zeros(md, np.float32)
            for _, sc, _ in must[2:]:
                oth = (oth + sc).astype(np.float32)
            score = (score + oth).astype(np.float32)
        if should:
            score = (score + opt).astype(np.float32)
            if msm:
                match &= n_hit >= msm
        cost = must[0][2]
    elif should:
        match, score = n_hit >= max(1, msm), opt
        cost = sum(c for _, _, c in should)
    else:
        return None
    for hit, _, _ in mustnot:
        match = match & ~hit
    return match, np.where(match, score, np.float32(0)).astype(np.float32), cost


def _is_phrase(t):
    return isinstance(t, tuple) and len(t) >= 2 and t[0] == "ph"


def tree_general(clauses, min_should_match=0):
    """The clause-list form tree_match_all takes (tests/tree_shapes.py) as a GENERAL tree: a node is a term id,
    ("ph", [term ids][, offsets]) or ("bool", [(occur, node), ...], minimum_number_should_match).  A member that is a
    list of term ids is a nested intersection (a BooleanQuery of Must terms), ("any", [term ids]) a nested union."""
    def member(t):
        if _is_phrase(t):
            return t
        if isinstance(t, tuple) and len(t) == 2 and t[0] == "any":
            return ("bool", [(SHOULD, x) for x in t[1]], 0)
        if isinstance(t, tuple) and len(t) >= 2 and t[0] == "bool":
            return t
        if isinstance(t, (list, tuple)):
            return ("bool", [(MUST, x) for x in t], 0)
        return t

    top = []
    for cl in clauses:
        if _is_phrase(cl[1]) or not isinstance(cl[1], (list, tuple)) or (isinstance(cl[1], tuple) and cl[1][0] in ("bool", "any")):
            top.append((cl[0], member(cl[1])))
        else:
            top.append((cl[0], ("bool", [(o, member(t)) for o, t in cl[1]], cl[2] if len(cl) > 2 else 0)))
    return ("bool", top, min_should_match)


def tree_match_all_general(seg, node):
    """BooleanWeight::complex_scorer applied recursively (boolean_weight.rs:225-233,236-431) in numpy over dense
    (match, score) arrays: every level through _combine_scorers.  Returns (docs ascending, f32 scores)."""
    md = seg.max_doc
    per = {}

    def leaf(t):
        if t not in per:
            hit = np.zeros(md, bool)
            sc = np.zeros(md, np.float32)
            if seg.terms[t].doc_freq:
                d, s = match_all(seg, [t], MODE_OR)
                hit[d] = True
                sc[d] = s
            per[t] = (hit, sc, seg.terms[t].doc_freq)
        return per[t]

    def phrase(terms, offsets=None):
        # PhraseScorer (phrase_scorer.rs:347-587): docs where the terms line up, bm25(sum of idfs, norm, phrase count);
        # cost = size_hint of the intersection of its lists * 10 * terms (:566-573, size_hint.rs:11-36)
        hit = np.zeros(md, bool)
        sc = np.zeros(md, np.float32)
        if all(seg.terms[t].doc_freq for t in terms):
            d, s = match_all(seg, list(terms), MODE_PHRASE, phrase_offsets=offsets)
            hit[d] = True
            sc[d] = s
        est, smallest, f = 0.0, 0.0, 1.3
        for i, t in enumerate(terms):
            df = float(seg.terms[t].doc_freq)
            if i == 0:
                est = smallest = df
            else:
                f = max(1.0, f - 0.1)
                est *= df / md * f
                smallest = min(smallest, df)
        return hit, sc, int(min(round(est), smallest)) * 10 * len(terms)

    def ev(n):
        if _is_phrase(n):
            return phrase(*n[1:])
        if isinstance(n, tuple) and n[0] == "bool":
            return _combine_scorers(md, [(o, ev(c)) for o, c in n[1]], n[2] if len(n) > 2 else 0)
        return leaf(n)

    res = ev(node)
    if res is None:
        return np.zeros(0, np.uint32), np.zeros(0, np.float32)
    docs = np.nonzero(res[0])[0]
    return docs.astype(np.uint32), res[1][docs]


def tree_match_all(seg, clauses, min_should_match=0):
    """A BooleanQuery whose clauses are terms, phrases or BooleanQuerys (tests/tree_shapes.py's clause-list form),
    restated from BooleanWeight::complex_scorer applied on every level (boolean_weight.rs:236-431; a nested query's
    scorer is just another Box<dyn Scorer> of its parent, :225-233).  clauses = [(occur, term_id) |
    (occur, [(inner occur, term_id | [term ids of a nested intersection] | ("any", [term ids of a nested union])), ...],
    nested minimum_number_should_match)]; a term_id can also be ("ph", [term ids][, offsets]): a PhraseQuery.  Every
    term scores with its own Bm25Weight.  Returns (docs ascending, f32 scores).  Sums of 3+ terms compare within
    1e-5 (the reference's own order follows scorer removal / cursor order)."""
    return tree_match_all_general(seg, tree_general(clauses, min_should_match))


def tree_match_all_c(seg, node):
    """The same tree through the C transliteration of the scorer tree (to_query.c: gs_build_node -> gs_complex:
    Intersection / BufferedUnionScorer / Disjunction / RequiredOptionalScorer / Exclude, PhraseScorer as a docset),
    driven like for_each_pruning_scorer.  node: the general form (tree_general)."""
    nodes, terms, weights, offs = [], [], [], []

    def add_term(t, w):
        terms.append(seg.terms[t])
        weights.append(w)
        offs.append(0)
        return len(terms) - 1

    def walk(n, occur):
        if _is_phrase(n):
            tl = list(n[1])
            ol = list(n[2]) if len(n) > 2 and n[2] is not None else list(range(len(tl)))
            w = default_weights(seg, tl, MODE_PHRASE)[0]
            first = len(terms)
            for t, o in zip(tl, ol):
                i = add_term(t, w)
                offs[i] = o
            nodes.append(TreeNode(2, occur, len(tl), first, 0))
        elif isinstance(n, tuple) and n[0] == "bool":
            at = len(nodes)
            nodes.append(TreeNode(1, occur, len(n[1]), 0, int(n[2]) if len(n) > 2 else 0))
            for o, c in n[1]:
                walk(c, int(o))
            assert nodes[at].n_kids == len(n[1])
        else:
            w = default_weights(seg, [n], MODE_OR)[0]
            nodes.append(TreeNode(0, occur, 0, add_term(n, w), 0))

    walk(node, MUST)
    na = (TreeNode * len(nodes))(*nodes)
    ta = (TermInfo * max(1, len(terms)))(*terms)
    wa = (Bm25 * max(1, len(weights)))(*weights)
    oa = (C.c_uint32 * max(1, len(offs)))(*offs)
    cap = max(1, seg.max_doc)
    docs = np.zeros(cap, np.uint32)
    scores = np.zeros(cap, np.float32)
    n = lib().to_tree_match_all(C.byref(seg.view), na, len(nodes), ta, wa, oa, len(terms), _u32(docs),
                                scores.ctypes.data_as(C.POINTER(C.c_float)), cap)
    assert n != (1 << 64) - 1, "malformed tree"
    return docs[:n], scores[:n]


def tree_search(seg, clauses, k, min_should_match=0, deleted=None):
    """Top-k of tree_match_all by (score desc, doc asc): [(score, doc)]."""
    d, s = tree_match_all(seg, clauses, min_should_match)
    if deleted is not None and len(d):
        keep = ~np.isin(d, deleted)
        d, s = d[keep], s[keep]
    order = np.lexsort((d, -s.astype(np.float64)))[:k]
    return [(float(s[i]), int(d[i])) for i in order]


def bool_spec(seg, term_ids, occurs, clause_of=None, min_should_match=0, k=1, boosts=None,
              total_num_docs=None, total_num_tokens=None, dfs=None):
    """QuerySpec of a boolean query for the C executor restatement (generic scorer tree); with the
    index-wide statistics (total_num_docs / total_num_tokens / doc freqs over all segments) the
    Bm25Weights are the ones Searcher hands every segment (bm25.rs:27-50)."""
    ws = default_weights(seg, term_ids, MODE_OR, total_num_docs, total_num_tokens, dfs, boosts=boosts)
    return QuerySpec(seg, term_ids, ws, MODE_BOOL, k, None, occurs, clause_of, min_should_match)


def bool_search(seg, term_ids, occurs, k, clause_of=None, min_should_match=0, boosts=None):
    """Top-k through the restated scorer tree (Intersection / BufferedUnionScorer / Disjunction /
    RequiredOptionalScorer / Exclude under for_each_pruning_scorer): [(score, doc)] sorted."""
    spec = bool_spec(seg, term_ids, occurs, clause_of, min_should_match, k, boosts)
    out = (Hit * max(1, k))()
    n = lib().to_search_exhaustive(C.byref(seg.view), C.byref(spec.q), out)
    lib().to_sort_hits(out, n)
    return _hits(out, n)


def bool_match_all_c(seg, term_ids, occurs, clause_of=None, min_should_match=0):
    """Every match of the restated scorer tree: (docs ascending, f32 scores)."""
    spec = bool_spec(seg, term_ids, occurs, clause_of, min_should_match)
    cap = max(1, seg.max_doc)
    docs = np.zeros(cap, np.uint32)
    scores = np.zeros(cap, np.float32)
    n = lib().to_match_all(C.byref(seg.view), C.byref(spec.q), _u32(docs),
                           scores.ctypes.data_as(C.POINTER(C.c_float)), cap)
    return docs[:n], scores[:n]


def decode_postings(seg, term_id):
    t = seg.terms[term_id]
    docs = np.zeros(max(1, t.doc_freq), np.uint32)
    tfs = np.zeros(max(1, t.doc_freq), np.uint32)
    n = lib().to_decode_postings(C.byref(seg.view), C.byref(t), _u32(docs), _u32(tfs))
    return docs[:n], tfs[:n]


def decode_positions(seg, term_id, cap):
    t = seg.terms[term_id]
    out = np.zeros(max(1, cap), np.uint32)
    n = lib().to_decode_positions(C.byref(seg.view), C.byref(t), _u32(out), cap)
    return out[: min(n, cap)], n


def merge_top_k(hits, offset, limit):
    """hits: iterable of (score, segment_ord, doc)."""
    hits = list(hits)
    arr = (GlobalHit * max(1, len(hits)))(*[GlobalHit(s, o, d) for (s, o, d) in hits])
    out = (GlobalHit * max(1, limit))()
    n = lib().to_merge_top_k(arr, len(hits), offset, limit, out)
    return [(float(out[i].score), int(out[i].segment_ord), int(out[i].doc)) for i in range(n)]


def zipf_queries(n_queries, n_terms_per_query, max_rank, seed):
    """Distinct Zipf(s=1)-distributed term ranks (0-based ids) per query."""
    r = Rng()
    L = lib()
    L.to_rng_seed(C.byref(r), seed)
    out = np.zeros((n_queries, n_terms_per_query), dtype=np.uint32)
    for i in range(n_queries):
        chosen = []
        while len(chosen) < n_terms_per_query:
            x = L.to_zipf_rank(C.byref(r), max_rank) - 1
            if x not in chosen:
                chosen.append(x)
        out[i] = chosen
    return out


def baseline_run(seg, specs, n_threads, want_hits=False):
    """Time to_search_pruned over QuerySpecs with query-level thread parallelism.
    Returns (wall_seconds, latencies[np.float64], hits or None)."""
    n = len(specs)
    qs = (Query * n)(*[s.q for s in specs])
    lat = np.zeros(n, np.float64)
    k = specs[0].q.k if n else 1
    hits = (Hit * (n * k))() if want_hits else None
    counts = np.zeros(n, np.uint32)
    wall = lib().to_baseline_run(C.byref(seg.view), qs, n, int(n_threads),
                                 lat.ctypes.data_as(C.POINTER(C.c_double)), hits, _u32(counts))
    res = None
    if want_hits:
        res = [_hits(hits[i * k:(i + 1) * k], int(counts[i])) for i in range(n)]
    return wall, lat, res
