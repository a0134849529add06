"""ctypes binding of libtantivy_amd.so: the raw C ABI (tq_*) and the C++ host mirror (tqh_*)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtantivy_amd.so")
if os.environ.get("TQ_LIB_PATH"):  # experiments: a variant build (tools/build_variant.py)
    LIB_PATH = os.environ["TQ_LIB_PATH"]

TERMINATED = 0x7FFFFFFF
TERM_ABSENT = 0xFFFFFFFF
MODE_AND, MODE_OR, MODE_PHRASE, MODE_BOOL, MODE_TERM = 0, 1, 2, 3, 4  # enum tq_mode + TQH_MODE_TERM
SHOULD, MUST, MUST_NOT = 0, 1, 2  # src/query/occur.rs
BASIC, WITH_FREQS, WITH_FREQS_AND_POSITIONS = 0, 1, 2


class TantivyAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tantivy_amd error %d: %s" % (code, msg))
        self.code = code


class TqQuery(C.Structure):
    _fields_ = [("n_terms", C.c_uint32), ("terms", C.POINTER(C.c_uint32)),
                ("weights", C.POINTER(C.c_float)), ("tf_cache", C.POINTER(C.c_float)),
                ("mode", C.c_uint8), ("phrase_offsets", C.POINTER(C.c_uint32)),
                ("k", C.c_uint32), ("occurs", C.POINTER(C.c_uint8)),
                ("clause_of", C.POINTER(C.c_uint8)), ("min_should_match", C.c_uint32),
                ("nested_occurs", C.POINTER(C.c_uint8)), ("clause_min_should", C.POINTER(C.c_uint8)),
                ("atom_of", C.POINTER(C.c_uint8))]


class TqSearchOpts(C.Structure):
    _fields_ = [("exhaustive", C.c_int32), ("bound_slack_ppm", C.c_uint32)]


OPT_DEFAULT = 0xFFFFFFFF


class TqBatchStats(C.Structure):
    _fields_ = [("algorithmic_bytes", C.c_uint64), ("matches", C.c_uint64),
                ("kernel_ms", C.c_float), ("total_ms", C.c_float), ("tiles", C.c_uint32),
                ("chunks", C.c_uint32), ("batches_averaged", C.c_uint32),
                ("host_plan_ms", C.c_float), ("kernel_mask", C.c_uint32), ("unique_bytes", C.c_uint64)]


# scan-kernel families (tq_batch_stats.kernel_mask, include/tantivy_amd.h)
KERNEL_AND_DENSE, KERNEL_AND, KERNEL_UNION, KERNEL_OR_WINDOWS, KERNEL_PHRASE = 0x1, 0x2, 0x4, 0x8, 0x10
KERNEL_PHRASE_SWEEP, KERNEL_BOOL, KERNEL_USHARE, KERNEL_XUNION, KERNEL_ASHARE = 0x20, 0x40, 0x80, 0x100, 0x200
KERNEL_BSHARE = 0x400
KERNEL_COUNT_BITMAPS = 0x800
KERNEL_TREE = 0x1000
NESTED_PHRASE = 0x10  # tq_query.nested_occurs flag: the atom is a PhraseQuery (include/tantivy_amd.h)
KERNEL_NAMES = {0x1: "and_dense", 0x2: "and", 0x4: "union", 0x8: "or_windows", 0x10: "phrase", 0x20: "phrase_sweep",
                0x40: "bool", 0x80: "ushare", 0x100: "xunion", 0x200: "ashare", 0x400: "bshare", 0x800: "count_bitmaps",
                0x1000: "tree"}


def kernel_names(mask):
    return [n for b, n in sorted(KERNEL_NAMES.items()) if mask & b]


class TqSubmitStats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("queries", C.c_uint64), ("max_batch", C.c_uint64)]


class TqSegmentStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "index_bytes", "positions_bytes", "fieldnorm_bytes", "alive_bytes", "term_table_bytes",
        "bitmap_bytes", "docmat_bytes", "posdir_bytes", "scratch_bytes", "dense_budget_bytes")] + [
        (n, C.c_uint32) for n in ("n_terms", "n_dense_lists", "n_docmat_columns", "probe_evictions")] + [
        ("device_scratch_bytes", C.c_uint64)]


class TqhTermInfo(C.Structure):
    _fields_ = [("term_id", C.c_uint32), ("doc_freq", C.c_uint32), ("postings_start", C.c_uint64),
                ("postings_end", C.c_uint64), ("positions_start", C.c_uint64),
                ("positions_end", C.c_uint64)]


class TqhQuery(C.Structure):
    _fields_ = [("mode", C.c_uint8), ("n_terms", C.c_uint32), ("terms", C.POINTER(C.c_uint32)),
                ("phrase_offsets", C.POINTER(C.c_uint32)), ("occurs", C.POINTER(C.c_uint8)),
                ("clause_of", C.POINTER(C.c_uint8)), ("min_should_match", C.c_uint32),
                ("boosts", C.POINTER(C.c_float)), ("nested_occurs", C.POINTER(C.c_uint8)),
                ("clause_min_should", C.POINTER(C.c_uint8)), ("atom_of", C.POINTER(C.c_uint8))]


_lib = None

EXPORTS = [
    "tq_init", "tq_shutdown", "tq_last_error", "tq_segment_upload", "tq_segment_upload_device",
    "tq_segment_free",
    "tq_term_prepare", "tq_search_batch", "tq_search_batch_device", "tq_search_batch_opts",
    "tq_search_batch_device_opts", "tq_merge_topk",
    "tq_merge_topk_device", "tq_copy_to_host_async", "tq_decode_postings", "tq_decode_position_deltas",
    "tq_last_batch_stats", "tq_segment_get_stats", "tq_segment_reserve_columns", "tq_set_option", "tq_segment_set_alive_bitset", "tq_count_batch",
    "tq_last_batch_match_counts", "tq_last_batch_query_kernels", "tq_encoder_create", "tq_encoder_free", "tq_encode_postings",
    "tq_encode_positions", "tq_encode_postings_device", "tq_encode_positions_device",
    "tq_encoder_last_kernel_ms", "tq_comm_unique_id", "tq_comm_init", "tq_comm_free",
    "tq_comm_info", "tq_allgather_topk",
    "tqh_last_error", "tqh_searcher_new", "tqh_searcher_free", "tqh_searcher_add_segment",
    "tqh_prepare_batch", "tqh_prepare_batch_next", "tqh_commit_next", "tqh_search_prepared", "tqh_collect_segment_prepared",
    "tqh_collect_segment_prepared_device", "tqh_searcher_add_remote_stats",
    "tqh_bm25_for_terms", "tqh_segment_raw", "tqh_term_handle", "tqh_term_dictionary_values",
    "tqh_term_info_store_open", "tqh_term_info_store_free", "tqh_term_info_store_num_terms",
    "tqh_term_info_store_get", "tqh_term_info_store_write", "tqh_searcher_add_segment_with_store",
    "tqh_count_prepared", "tqh_searcher_add_segment_device_with_store",
]


def lib():
    """Loads the HIP library.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TantivyAmdError(-1, "%s not built (run `python -m tantivy_amd.build` or "
                                  "__graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u8p, u32p, f32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.tq_last_error.restype = C.c_char_p
    L.tqh_last_error.restype = C.c_char_p
    L.tq_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.tq_shutdown.argtypes = [vp]
    L.tq_segment_upload.argtypes = [vp, C.c_int, C.c_uint32, vp, C.c_size_t, vp, C.c_size_t, vp,
                                    C.c_size_t, C.c_uint8, C.POINTER(vp)]
    L.tq_segment_upload_device.argtypes = [vp, C.c_int, C.c_uint32, vp, C.c_size_t, vp, C.c_size_t, vp,
                                           C.c_size_t, C.c_uint8, C.POINTER(vp)]
    L.tq_segment_free.argtypes = [vp]
    L.tq_term_prepare.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                  u32p]
    L.tq_search_batch.argtypes = [vp, C.POINTER(TqQuery), C.c_uint32, C.c_uint32, f32p, u32p, u32p]
    L.tq_search_batch_device.argtypes = [vp, C.POINTER(TqQuery), C.c_uint32, C.c_uint32, vp, vp,
                                         vp, vp]
    L.tq_search_batch_opts.argtypes = [vp, C.POINTER(TqQuery), C.c_uint32, C.c_uint32, f32p, u32p,
                                       u32p, C.POINTER(TqSearchOpts)]
    L.tq_search_batch_device_opts.argtypes = [vp, C.POINTER(TqQuery), C.c_uint32, C.c_uint32, vp,
                                              vp, vp, C.POINTER(TqSearchOpts), vp]
    L.tq_merge_topk.argtypes = [f32p, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, f32p, u32p, u32p, u32p]
    L.tq_copy_to_host_async.argtypes = [vp, C.c_int, vp, vp, C.c_size_t, vp]
    L.tq_merge_topk_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    L.tq_decode_postings.argtypes = [vp, C.c_uint32, u32p, u32p]
    L.tq_decode_position_deltas.argtypes = [vp, C.c_uint32, u32p, C.c_uint64,
                                            C.POINTER(C.c_uint64)]
    L.tq_last_batch_stats.argtypes = [vp, C.POINTER(TqBatchStats)]
    L.tq_segment_get_stats.argtypes = [vp, C.POINTER(TqSegmentStats)]
    L.tq_segment_reserve_columns.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint32]
    L.tq_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.tq_segment_set_alive_bitset.argtypes = [vp, vp, C.c_size_t]
    L.tq_count_batch.argtypes = [vp, C.POINTER(TqQuery), C.c_uint32, u32p]
    L.tq_last_batch_match_counts.argtypes = [vp, u32p, C.c_uint32]
    L.tq_last_batch_query_kernels.argtypes = [vp, u32p, C.c_uint32]
    u64p = C.POINTER(C.c_uint64)
    L.tq_encoder_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.tq_encoder_free.argtypes = [vp]
    L.tq_encoder_free.restype = None
    L.tq_encode_postings.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.c_float,
                                     C.c_uint8, vp, C.c_uint64, vp, u64p]
    L.tq_encode_positions.argtypes = [vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, u64p]
    L.tq_encode_postings_device.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32,
                                            C.c_float, C.c_uint8, vp, C.c_uint64, vp, u64p, vp]
    L.tq_encode_positions_device.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint64, vp, u64p,
                                             vp]
    L.tq_encoder_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.tq_comm_unique_id.argtypes = [vp]
    L.tq_comm_init.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.tq_comm_free.argtypes = [vp]
    L.tq_comm_free.restype = None
    L.tq_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    L.tq_allgather_topk.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.tqh_searcher_new.argtypes = [vp, C.POINTER(vp)]
    L.tqh_searcher_free.argtypes = [vp]
    L.tqh_searcher_add_segment.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint8, vp, C.c_size_t, vp,
                                           C.c_size_t, vp, C.c_size_t, C.POINTER(TqhTermInfo),
                                           C.c_uint32]
    L.tqh_prepare_batch.argtypes = [vp, C.POINTER(TqhQuery), C.c_uint32]
    L.tqh_prepare_batch_next.argtypes = [vp, C.POINTER(TqhQuery), C.c_uint32]
    L.tqh_commit_next.argtypes = [vp]
    L.tqh_search_prepared.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, u32p, u32p, u32p]
    L.tqh_search_concurrent.argtypes = [vp, C.POINTER(TqhQuery), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        f32p, u32p, u32p, u32p, f32p, C.POINTER(C.c_double)]
    L.tq_get_submit_stats.argtypes = [vp, C.POINTER(TqSubmitStats), C.c_int]
    L.tq_search_one.argtypes = [vp, C.POINTER(TqQuery), C.POINTER(TqSearchOpts), f32p, u32p, u32p]
    L.tq_submit.argtypes = [vp, C.POINTER(TqQuery), C.POINTER(TqSearchOpts), f32p, u32p, u32p, C.POINTER(vp)]
    L.tq_wait.argtypes = [vp]
    L.tqh_collect_segment_prepared.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, u32p, u32p]
    L.tqh_collect_segment_prepared_device.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.tqh_searcher_add_remote_stats.argtypes = [vp, C.c_uint64, C.c_uint64, u32p, u32p, C.c_uint32]
    L.tqh_bm25_for_terms.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint64, C.c_uint64,
                                     C.c_float, f32p, f32p]
    L.tqh_segment_raw.restype = vp
    L.tqh_segment_raw.argtypes = [vp, C.c_uint32]
    L.tqh_term_handle.restype = C.c_uint32
    L.tqh_term_handle.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.tqh_term_dictionary_values.argtypes = [vp, C.c_size_t, u64p, u64p]
    L.tqh_count_prepared.argtypes = [vp, u64p]
    L.tqh_term_info_store_open.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.tqh_term_info_store_free.argtypes = [vp]
    L.tqh_term_info_store_free.restype = None
    L.tqh_term_info_store_num_terms.argtypes = [vp]
    L.tqh_term_info_store_num_terms.restype = C.c_uint64
    L.tqh_term_info_store_get.argtypes = [vp, vp, C.c_uint32, C.POINTER(TqhTermInfo)]
    L.tqh_term_info_store_write.argtypes = [C.POINTER(TqhTermInfo), C.c_uint32, vp, C.c_uint64, u64p]
    L.tqh_searcher_add_segment_with_store.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint8, vp,
                                                      C.c_size_t, vp, C.c_size_t, vp, C.c_size_t,
                                                      vp, C.c_size_t]
    L.tqh_searcher_add_segment_device_with_store.argtypes = [
        vp, C.c_int, C.c_uint32, C.c_uint8, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t,
        C.c_uint64, vp, C.c_size_t]
    _lib = L
    return L


def _check(rc, host=False):
    if rc != 0:
        L = lib()
        msg = (L.tqh_last_error() if host else L.tq_last_error()) or b""
        if host and not msg:
            msg = L.tq_last_error() or b""
        raise TantivyAmdError(rc, msg.decode("utf-8", "replace"))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def bm25_for_terms(doc_freqs, total_num_docs, total_num_tokens, boost=1.0):
    """Product-side Bm25Weight::for_terms: returns (weight, cache[256])."""
    dfs = (C.c_uint64 * len(doc_freqs))(*[int(d) for d in doc_freqs])
    w = C.c_float()
    cache = np.zeros(256, np.float32)
    _check(lib().tqh_bm25_for_terms(dfs, len(doc_freqs), int(total_num_docs), int(total_num_tokens),
                                    C.c_float(boost), C.byref(w), _f32(cache)), host=True)
    return float(w.value), cache


def term_dictionary_values(field_file):
    """(offset, length) of the TermInfoStore inside a field's term dictionary file."""
    b = np.frombuffer(bytes(field_file), dtype=np.uint8)
    off, ln = C.c_uint64(), C.c_uint64()
    _check(lib().tqh_term_dictionary_values(b.ctypes.data, b.size, C.byref(off), C.byref(ln)), host=True)
    return int(off.value), int(ln.value)


class TermInfoStore:
    """Host-side TermInfoStore (tantivy_amd/host/term_info_store.hpp)."""

    def __init__(self, store_bytes):
        b = np.frombuffer(bytes(store_bytes), dtype=np.uint8)
        self._h = C.c_void_p()
        _check(lib().tqh_term_info_store_open(b.ctypes.data, b.size, C.byref(self._h)), host=True)

    def num_terms(self):
        return int(lib().tqh_term_info_store_num_terms(self._h))

    def get(self, ords):
        """[(doc_freq, postings_start, postings_end, positions_start, positions_end), ...]"""
        o = np.ascontiguousarray(ords, np.uint64)
        out = (TqhTermInfo * max(1, o.size))()
        _check(lib().tqh_term_info_store_get(self._h, o.ctypes.data, o.size, out), host=True)
        return [(t.doc_freq, t.postings_start, t.postings_end, t.positions_start, t.positions_end)
                for t in out[: o.size]]

    def close(self):
        if self._h:
            lib().tqh_term_info_store_free(self._h)
            self._h = None

    @staticmethod
    def serialize(term_infos):
        """TermInfoStoreWriter over (doc_freq, ps, pe, qs, qe) tuples in ordinal order."""
        n = len(term_infos)
        tis = (TqhTermInfo * max(1, n))()
        for i, t in enumerate(term_infos):
            tis[i] = TqhTermInfo(i, *[int(x) for x in t])
        need = C.c_uint64()
        out = np.zeros(64 + 48 * n, np.uint8)
        rc = lib().tqh_term_info_store_write(tis, n, out.ctypes.data, out.size, C.byref(need))
        if rc != 0 and need.value > out.size:
            out = np.zeros(need.value, np.uint8)
            rc = lib().tqh_term_info_store_write(tis, n, out.ctypes.data, out.size, C.byref(need))
        _check(rc, host=True)
        return out[: need.value].tobytes()


class Encoder:
    """Device-side PostingsSerializer / PositionSerializer (tq_encode_*), host buffers in and out."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        arr = (C.c_int * 1)(device)
        _check(lib().tq_init(arr, 1, C.byref(self._ctx)))
        self._e = C.c_void_p()
        _check(lib().tq_encoder_create(self._ctx, device, C.byref(self._e)))

    def close(self):
        if self._e:
            lib().tq_encoder_free(self._e)
            self._e = None
        if self._ctx:
            lib().tq_shutdown(self._ctx)
            self._ctx = None

    def _run(self, call, n_terms, est):
        out = np.zeros(max(16, est), np.uint8)
        ots = np.zeros(n_terms + 1, np.uint64)
        need = C.c_uint64()
        rc = call(out, ots, need)
        if rc != 0 and need.value > out.size:  # too small: the call reported the size
            out = np.zeros(need.value, np.uint8)
            rc = call(out, ots, need)
        _check(rc)
        return out[: need.value].copy(), ots

    def encode_postings(self, term_starts, docs, tfs, fieldnorm_ids, num_docs, avg_fieldnorm,
                        record_option, out_cap=None):
        ts = np.ascontiguousarray(term_starts, np.uint64)
        docs = np.ascontiguousarray(docs, np.uint32)
        tfs = None if tfs is None else np.ascontiguousarray(tfs, np.uint32)
        fn = None if fieldnorm_ids is None else np.ascontiguousarray(fieldnorm_ids, np.uint8)

        def call(out, ots, need):
            return lib().tq_encode_postings(
                self._e, len(ts) - 1, ts.ctypes.data, docs.ctypes.data,
                None if tfs is None else tfs.ctypes.data, None if fn is None else fn.ctypes.data,
                int(num_docs), C.c_float(avg_fieldnorm), int(record_option), out.ctypes.data,
                out.size, ots.ctypes.data, C.byref(need))

        est = out_cap if out_cap is not None else int(docs.size) * 3 + 64 * len(ts)
        return self._run(call, len(ts) - 1, est)

    def encode_positions(self, term_starts, deltas, out_cap=None):
        ts = np.ascontiguousarray(term_starts, np.uint64)
        deltas = np.ascontiguousarray(deltas, np.uint32)

        def call(out, ots, need):
            return lib().tq_encode_positions(self._e, len(ts) - 1, ts.ctypes.data,
                                             deltas.ctypes.data, out.ctypes.data, out.size,
                                             ots.ctypes.data, C.byref(need))

        est = out_cap if out_cap is not None else int(deltas.size) * 2 + 64 * len(ts)
        return self._run(call, len(ts) - 1, est)

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().tq_encoder_last_kernel_ms(self._e, C.byref(ms)))
        return float(ms.value)

    @property
    def raw(self):
        return self._e


class DeviceIndex:
    """A Searcher over device-resident segments (C++ host mirror, tantivy_amd/host/searcher.hpp).

    segments: iterable of objects with max_doc, record_option, idx (np.uint8, header included),
    pos, fieldnorm (or None) and terms: list of (doc_freq, postings_start, postings_end,
    positions_start, positions_end); term id = list index.
    """

    def __init__(self, segments=(), devices=None):
        L = lib()
        self._ctx = C.c_void_p()
        devs = list(devices) if devices is not None else [0]
        arr = (C.c_int * len(devs))(*devs)
        _check(L.tq_init(arr, len(devs), C.byref(self._ctx)))
        self._s = C.c_void_p()
        _check(L.tqh_searcher_new(self._ctx, C.byref(self._s)), host=True)
        self._devs = devs
        self.n_segments = 0
        self._keep = []
        self._n_prepared = 0
        for i, seg in enumerate(segments):
            self.add_segment(seg, devs[i % len(devs)])

    def add_segment(self, seg, device=0, term_info_store=None):
        """term_info_store: the bytes of the segment's TermInfoStore; then term ids are term
        ordinals and seg.terms is not consulted (TermInfos are decoded by the library)."""
        L = lib()
        idx = np.ascontiguousarray(seg.idx[: seg.idx_len] if hasattr(seg, "idx_len") else seg.idx,
                                   dtype=np.uint8)
        pos_len = getattr(seg, "pos_len", len(seg.pos) if seg.pos is not None else 0)
        pos = np.ascontiguousarray(seg.pos[:pos_len], dtype=np.uint8) if pos_len else None
        fn = None if seg.fieldnorm is None else np.ascontiguousarray(seg.fieldnorm, dtype=np.uint8)
        if term_info_store is not None:
            st = np.frombuffer(bytes(term_info_store), dtype=np.uint8)
            _check(L.tqh_searcher_add_segment_with_store(
                self._s, int(device), int(seg.max_doc), int(seg.record_option),
                idx.ctypes.data, idx.size, pos.ctypes.data if pos is not None else None,
                pos.size if pos is not None else 0, fn.ctypes.data if fn is not None else None,
                fn.size if fn is not None else 0, st.ctypes.data, st.size), host=True)
            self.n_segments += 1
            return
        n = len(seg.terms)
        tis = (TqhTermInfo * max(1, n))()
        for i, t in enumerate(seg.terms):
            if isinstance(t, (tuple, list)):
                df, ps, pe, qs, qe = t
            else:
                df, ps, pe, qs, qe = (t.doc_freq, t.postings_start, t.postings_end,
                                      t.positions_start, t.positions_end)
            tis[i] = TqhTermInfo(i, int(df), int(ps), int(pe), int(qs), int(qe))
        _check(L.tqh_searcher_add_segment(
            self._s, int(device), int(seg.max_doc), int(seg.record_option),
            idx.ctypes.data, idx.size, pos.ctypes.data if pos is not None else None,
            pos.size if pos is not None else 0, fn.ctypes.data if fn is not None else None,
            fn.size if fn is not None else 0, tis, n), host=True)
        self.n_segments += 1

    def add_segment_device(self, max_doc, record_option, d_idx, d_pos, d_fieldnorm,
                           total_num_tokens, term_info_store, device=0):
        """Segment whose sub-files are torch uint8 tensors on the GPU (d_idx with the 8-byte
        header; d_pos / d_fieldnorm may be None); term ids = ordinals of term_info_store."""
        st = np.frombuffer(bytes(term_info_store), dtype=np.uint8)
        _check(lib().tqh_searcher_add_segment_device_with_store(
            self._s, int(device), int(max_doc), int(record_option), d_idx.data_ptr(), d_idx.numel(),
            d_pos.data_ptr() if d_pos is not None else None,
            d_pos.numel() if d_pos is not None else 0,
            d_fieldnorm.data_ptr() if d_fieldnorm is not None else None,
            d_fieldnorm.numel() if d_fieldnorm is not None else 0, int(total_num_tokens),
            st.ctypes.data, st.size), host=True)
        self.n_segments += 1

    def add_remote_stats(self, max_doc, total_num_tokens, doc_freqs):
        """doc_freqs: sequence indexed by term id (segments held by other ranks)."""
        ids = np.arange(len(doc_freqs), dtype=np.uint32)
        dfs = np.ascontiguousarray(doc_freqs, dtype=np.uint32)
        _check(lib().tqh_searcher_add_remote_stats(self._s, int(max_doc), int(total_num_tokens),
                                                   _u32(ids), _u32(dfs), len(dfs)), host=True)

    def close(self):
        L = lib()
        if getattr(self, "_prep_pool", None) is not None:
            self._prep_pool.shutdown(wait=True)
            self._prep_pool = None
        if self._s:
            L.tqh_searcher_free(self._s)
            self._s = C.c_void_p()
        if self._ctx:
            L.tq_shutdown(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-mirror path (Query::weight + Searcher::search)
    def _host_queries(self, queries):
        """-> (tqh_query array, keep-alive list).  queries: list of (mode, [term ids]), (MODE_PHRASE, [term ids], [offsets]) or
        (MODE_BOOL, [term ids], [occurs][, clause_of | None[, min_should_match]]) with occurs in
        {SHOULD, MUST, MUST_NOT}; terms sharing a clause_of value form one nested union.  A trailing
        dict {"boosts": [...]} wraps every term query in BoostQuery(boost); {"nested_occurs": [...]}
        gives the occur of every term INSIDE its clause_of group (255 = Should: a nested union), e.g.
        `+a +(+b -c)` = (MODE_BOOL, [a, b, c], [MUST]*3, [0, 1, 1], 0, {"nested_occurs": [255, 1, 2]}).  A phrase
        inside a boolean query: its terms share an "atom_of" value, carry nested_occurs | NESTED_PHRASE and their
        "phrase_offsets" — `+"a b" +c` = (MODE_BOOL, [a, b, c], [MUST]*3, [0, 0, 1], 0, {"nested_occurs": [0x11, 0x11, 1],
        "atom_of": [0, 0, 0], "phrase_offsets": [0, 1, 0]})."""
        n = len(queries)
        qs = (TqhQuery * max(1, n))()
        keep = []
        for i, q in enumerate(queries):
            extra = q[-1] if isinstance(q[-1], dict) else None
            if extra is not None:
                q = q[:-1]
            mode, terms = q[0], q[1]
            if extra and extra.get("boosts") is not None:
                ba = (C.c_float * len(terms))(*[float(b) for b in extra["boosts"]])
                keep.append(ba)
                qs[i].boosts = C.cast(ba, C.POINTER(C.c_float))
            if extra and extra.get("nested_occurs") is not None:
                na = (C.c_uint8 * len(terms))(*[int(o) for o in extra["nested_occurs"]])
                keep.append(na)
                qs[i].nested_occurs = C.cast(na, C.POINTER(C.c_uint8))
            if extra and extra.get("atom_of") is not None:
                aa = (C.c_uint8 * len(terms))(*[int(o) for o in extra["atom_of"]])
                keep.append(aa)
                qs[i].atom_of = C.cast(aa, C.POINTER(C.c_uint8))
            if extra and extra.get("phrase_offsets") is not None:  # MODE_BOOL: offsets of the terms of phrases inside it
                pa = (C.c_uint32 * len(terms))(*[int(o) for o in extra["phrase_offsets"]])
                keep.append(pa)
                qs[i].phrase_offsets = C.cast(pa, C.POINTER(C.c_uint32))
            if extra and extra.get("clause_min_should") is not None:  # {clause_of value: nested minimum}
                ma = (C.c_uint8 * 16)(*[int(extra["clause_min_should"].get(c, 0)) for c in range(16)])
                keep.append(ma)
                qs[i].clause_min_should = C.cast(ma, C.POINTER(C.c_uint8))
            ta = (C.c_uint32 * len(terms))(*[int(t) for t in terms])
            keep.append(ta)
            qs[i].mode = mode
            qs[i].n_terms = len(terms)
            qs[i].terms = C.cast(ta, C.POINTER(C.c_uint32))
            if mode == MODE_BOOL:
                oc = (C.c_uint8 * len(terms))(*[int(o) for o in q[2]])
                keep.append(oc)
                qs[i].occurs = C.cast(oc, C.POINTER(C.c_uint8))
                if len(q) > 3 and q[3] is not None:
                    co = (C.c_uint8 * len(terms))(*[int(o) for o in q[3]])
                    keep.append(co)
                    qs[i].clause_of = C.cast(co, C.POINTER(C.c_uint8))
                if len(q) > 4:
                    qs[i].min_should_match = int(q[4])
            elif len(q) > 2 and q[2] is not None:
                oa = (C.c_uint32 * len(terms))(*[int(o) for o in q[2]])
                keep.append(oa)
                qs[i].phrase_offsets = C.cast(oa, C.POINTER(C.c_uint32))
        return qs, keep

    def prepare(self, queries):
        """Query::weight for a batch (see _host_queries for the query tuples)."""
        self.prepare_marshalled(self.marshal(queries))

    def marshal(self, queries):
        """The query tuples of a batch as the C structs tqh_prepare_batch takes (the Python side of a caller's
        request decoding: done ahead of a timed region that is to hold Query::weight itself)."""
        qs, keep = self._host_queries(queries)
        return qs, keep, len(queries)

    def prepare_marshalled(self, m):
        """Query::weight (+ tq_term_prepare and first-use tables of terms not seen before) for a marshalled batch."""
        qs, _, n = m
        _check(lib().tqh_prepare_batch(self._s, qs, n), host=True)
        self._n_prepared = n

    def prepare_next_async(self, m):
        """Query::weight of the NEXT batch on a helper thread (ctypes drops the GIL for the call) while this thread
        executes the current one; commit_next() waits for it and makes it the current batch.  Returns nothing: one
        batch can be in preparation at a time."""
        import concurrent.futures

        if getattr(self, "_prep_pool", None) is None:
            # the helper keeps off the caller's physical core: woken by the caller, the scheduler likes to put it on the
            # SMT sibling of the caller's CPU, where the two halve each other (Query::weight for 10 000 queries: 0.65 ms
            # alone, 2.5 ms on the sibling of a planning thread)
            avoid = set()
            try:
                cpu = os.sched_getcpu()
                with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % cpu) as f:
                    for part in f.read().strip().split(","):
                        lo, _, hi = part.partition("-")
                        avoid.update(range(int(lo), int(hi or lo) + 1))
            except (OSError, ValueError, AttributeError):
                avoid = set()

            def keep_off():
                try:
                    allowed = os.sched_getaffinity(0) - avoid
                    if allowed:
                        os.sched_setaffinity(0, allowed)
                except (OSError, AttributeError):
                    pass

            self._prep_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, initializer=keep_off)
        qs, _, n = m

        def work():
            import time as _t

            t0 = _t.perf_counter()
            _check(lib().tqh_prepare_batch_next(self._s, qs, n), host=True)
            return _t.perf_counter() - t0

        self._prep_next = (self._prep_pool.submit(work), n, m)

    def commit_next(self):
        """-> seconds the preparation took on its thread."""
        fut, n, _ = self._prep_next
        secs = fut.result()
        self._prep_next = None
        _check(lib().tqh_commit_next(self._s), host=True)
        self._n_prepared = n
        return secs

    def search_concurrent(self, queries, limit, n_threads, offset=0):
        """Searcher::search from n_threads host threads at once, one query per call (tantivy's own call
        pattern): the per-segment calls of concurrent threads are coalesced into batched launches
        (tq_search_one).  Returns (scores, segment_ords, docs, counts, latency_ms per query, wall_ms)."""
        qs, keep = self._host_queries(queries)
        n = len(queries)
        scores = np.zeros((n, limit), np.float32)
        ords = np.zeros((n, limit), np.uint32)
        docs = np.zeros((n, limit), np.uint32)
        counts = np.zeros(n, np.uint32)
        lat = np.zeros(max(1, n), np.float32)
        wall = C.c_double()
        _check(lib().tqh_search_concurrent(self._s, qs, n, int(offset), int(limit), int(n_threads),
                                           _f32(scores), _u32(ords), _u32(docs), _u32(counts), _f32(lat),
                                           C.byref(wall)), host=True)
        return scores, ords, docs, counts, lat[:n], float(wall.value)

    def submit_stats(self, segment_ord=0, reset=False):
        """Launches issued for queries that came through tq_submit / tq_search_one on one segment."""
        st = TqSubmitStats()
        _check(lib().tq_get_submit_stats(self.segment_raw(segment_ord), C.byref(st), 1 if reset else 0))
        return {"batches": int(st.batches), "queries": int(st.queries), "max_batch": int(st.max_batch)}

    def search_prepared(self, limit, offset=0):
        n = self._n_prepared
        scores = np.zeros((n, limit), np.float32)
        ords = np.zeros((n, limit), np.uint32)
        docs = np.zeros((n, limit), np.uint32)
        counts = np.zeros(n, np.uint32)
        _check(lib().tqh_search_prepared(self._s, int(offset), int(limit), _f32(scores),
                                         _u32(ords), _u32(docs), _u32(counts)), host=True)
        return scores, ords, docs, counts

    def collect_segment_prepared(self, segment_ord, k):
        n = self._n_prepared
        scores = np.zeros((n, k), np.float32)
        docs = np.zeros((n, k), np.uint32)
        counts = np.zeros(n, np.uint32)
        _check(lib().tqh_collect_segment_prepared(self._s, int(segment_ord), int(k), _f32(scores),
                                                  _u32(docs), _u32(counts)), host=True)
        return scores, docs, counts

    def collect_segment_prepared_device(self, segment_ord, k, d_scores, d_docs, d_counts,
                                        stream=None):
        """torch tensors (float32 [n,k], int32 [n,k], int32 [n]) on the segment's GPU."""
        _check(lib().tqh_collect_segment_prepared_device(
            self._s, int(segment_ord), int(k), d_scores.data_ptr(), d_docs.data_ptr(),
            d_counts.data_ptr(), C.c_void_p(stream) if stream else None), host=True)

    @property
    def ctx(self):
        return self._ctx

    def search(self, queries, limit, offset=0):
        self.prepare(queries)
        return self.search_prepared(limit, offset)

    def count(self, queries):
        """Searcher::search(&query, &Count) for a batch: alive matching docs over all segments."""
        self.prepare(queries)
        out = np.zeros(max(1, len(queries)), np.uint64)
        _check(lib().tqh_count_prepared(self._s, out.ctypes.data_as(C.POINTER(C.c_uint64))), host=True)
        return out[: len(queries)]

    # ---- raw C ABI access (parity tests)
    def segment_raw(self, segment_ord=0):
        return C.c_void_p(lib().tqh_segment_raw(self._s, segment_ord))

    def term_handle(self, term_id, segment_ord=0):
        return lib().tqh_term_handle(self._s, segment_ord, int(term_id))

    def decode_postings(self, term_id, doc_freq, segment_ord=0):
        h = self.term_handle(term_id, segment_ord)
        if h == TERM_ABSENT:
            raise TantivyAmdError(1, lib().tq_last_error().decode() or "term absent")
        docs = np.zeros(max(1, doc_freq), np.uint32)
        tfs = np.zeros(max(1, doc_freq), np.uint32)
        _check(lib().tq_decode_postings(self.segment_raw(segment_ord), h, _u32(docs), _u32(tfs)))
        return docs[:doc_freq], tfs[:doc_freq]

    def decode_position_deltas(self, term_id, cap, segment_ord=0):
        h = self.term_handle(term_id, segment_ord)
        out = np.zeros(max(1, cap), np.uint32)
        n = C.c_uint64()
        _check(lib().tq_decode_position_deltas(self.segment_raw(segment_ord), h, _u32(out), cap,
                                               C.byref(n)))
        return out[: min(cap, n.value)], n.value

    def set_alive_bitset(self, alive_bytes, segment_ord=0):
        """alive_bytes: the `.del` body (BitSet::serialize) as bytes / np.uint8, or None."""
        if alive_bytes is None:
            _check(lib().tq_segment_set_alive_bitset(self.segment_raw(segment_ord), None, 0))
            return
        a = np.ascontiguousarray(np.frombuffer(bytes(alive_bytes), dtype=np.uint8))
        _check(lib().tq_segment_set_alive_bitset(self.segment_raw(segment_ord), a.ctypes.data,
                                                 a.size))

    def last_batch_match_counts(self, n, segment_ord=0):
        out = np.zeros(max(1, n), np.uint32)
        _check(lib().tq_last_batch_match_counts(self.segment_raw(segment_ord), _u32(out), n))
        return out[:n]

    def last_batch_query_kernels(self, n, segment_ord=0):
        """TQ_KERNEL_* bit of every query of the last batch (option "record_query_kernels" set before it)."""
        out = np.zeros(max(1, n), np.uint32)
        _check(lib().tq_last_batch_query_kernels(self.segment_raw(segment_ord), _u32(out), n))
        return out[:n]

    def set_option(self, name, value, segment_ord=None):
        ords = range(self.n_segments) if segment_ord is None else [segment_ord]
        for o in ords:
            _check(lib().tq_set_option(self.segment_raw(o), name.encode(), int(value)))

    def last_batch_stats(self, segment_ord=0):
        st = TqBatchStats()
        _check(lib().tq_last_batch_stats(self.segment_raw(segment_ord), C.byref(st)))
        return {"algorithmic_bytes": st.algorithmic_bytes, "matches": st.matches,
                "kernel_ms": st.kernel_ms, "total_ms": st.total_ms, "tiles": st.tiles,
                "chunks": st.chunks, "batches_averaged": st.batches_averaged,
                "host_plan_ms": st.host_plan_ms, "kernel_mask": int(st.kernel_mask),
                "kernels": kernel_names(int(st.kernel_mask)), "unique_bytes": int(st.unique_bytes)}

    def segment_stats(self, segment_ord=0):
        """Resident HBM bytes of one segment by kind (tq_segment_get_stats)."""
        st = TqSegmentStats()
        _check(lib().tq_segment_get_stats(self.segment_raw(segment_ord), C.byref(st)))
        out = {n: int(getattr(st, n)) for n, _ in TqSegmentStats._fields_}
        out["derived_bytes"] = (out["term_table_bytes"] + out["bitmap_bytes"] + out["docmat_bytes"]
                                + out["posdir_bytes"])
        out["tantivy_bytes"] = (out["index_bytes"] + out["positions_bytes"] + out["fieldnorm_bytes"]
                                + out["alive_bytes"])
        return out

    def raw_count(self, queries, weights, cache, segment_ord=0):
        """Direct tq_count_batch (Count collector): alive matches per query."""
        n = len(queries)
        qs = (TqQuery * max(1, n))()
        keep = []
        cache = np.ascontiguousarray(cache, np.float32)
        for i, q in enumerate(queries):
            mode, terms = q[0], q[1]
            hs = (C.c_uint32 * len(terms))(*[self.term_handle(t, segment_ord) for t in terms])
            ws = (C.c_float * len(weights[i]))(*weights[i])
            keep += [hs, ws]
            qs[i].n_terms = len(terms)
            qs[i].terms = C.cast(hs, C.POINTER(C.c_uint32))
            qs[i].weights = C.cast(ws, C.POINTER(C.c_float))
            qs[i].tf_cache = _f32(cache)
            qs[i].mode = mode
            if mode == MODE_PHRASE:
                oa = (C.c_uint32 * len(terms))(*(q[2] if len(q) > 2 and q[2] is not None
                                                 else range(len(terms))))
                keep.append(oa)
                qs[i].phrase_offsets = C.cast(oa, C.POINTER(C.c_uint32))
            qs[i].k = 1
        counts = np.zeros(max(1, n), np.uint32)
        _check(lib().tq_count_batch(self.segment_raw(segment_ord), qs, n, _u32(counts)))
        return counts[:n]

    def raw_search(self, queries, weights, cache, k, segment_ord=0, stride=None, opts=None):
        """Direct tq_search_batch[_opts]: queries = list of (mode, [term ids], offsets|None);
        weights = list of per-query float lists; cache = np.float32[256]; opts = (exhaustive,
        bound_slack_ppm) for this call only (tq_search_opts) or None."""
        n = len(queries)
        stride = k if stride is None else stride
        qs = (TqQuery * max(1, n))()
        keep = []
        cache = np.ascontiguousarray(cache, np.float32)
        for i, q in enumerate(queries):
            mode, terms = q[0], q[1]
            hs = (C.c_uint32 * len(terms))(*[self.term_handle(t, segment_ord) for t in terms])
            ws = (C.c_float * len(weights[i]))(*weights[i])
            keep += [hs, ws]
            qs[i].n_terms = len(terms)
            qs[i].terms = C.cast(hs, C.POINTER(C.c_uint32))
            qs[i].weights = C.cast(ws, C.POINTER(C.c_float))
            qs[i].tf_cache = _f32(cache)
            qs[i].mode = mode
            if len(q) > 2 and q[2] is not None:
                oa = (C.c_uint32 * len(terms))(*q[2])
                keep.append(oa)
                qs[i].phrase_offsets = C.cast(oa, C.POINTER(C.c_uint32))
            if len(q) > 3 and q[3] is not None:  # TQ_MODE_BOOL: enum tq_occur
                oc = (C.c_uint8 * len(terms))(*q[3])
                keep.append(oc)
                qs[i].occurs = C.cast(oc, C.POINTER(C.c_uint8))
            if len(q) > 4 and q[4] is not None:
                co = (C.c_uint8 * len(terms))(*q[4])
                keep.append(co)
                qs[i].clause_of = C.cast(co, C.POINTER(C.c_uint8))
            if len(q) > 5:
                qs[i].min_should_match = int(q[5])
            qs[i].k = k
        scores = np.zeros((n, stride), np.float32)
        docs = np.zeros((n, stride), np.uint32)
        counts = np.zeros(n, np.uint32)
        if opts is not None:
            o = TqSearchOpts(int(opts[0]), int(opts[1]))
            _check(lib().tq_search_batch_opts(self.segment_raw(segment_ord), qs, n, stride,
                                              _f32(scores), _u32(docs), _u32(counts), C.byref(o)))
        else:
            _check(lib().tq_search_batch(self.segment_raw(segment_ord), qs, n, stride, _f32(scores),
                                         _u32(docs), _u32(counts)))
        return scores, docs, counts
