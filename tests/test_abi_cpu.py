"""CPU-only checks of the product library: it loads, exports every symbol the headers declare,
fails loudly without a GPU, and its host-side pieces (BM25 weights, merge_top_k) match the
oracle.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def B():
    from tantivy_amd import binding

    binding.lib()
    return binding


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tqh?_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(B):
    L = B.lib()
    names = _declared_functions("tantivy_amd.h") + _declared_functions("tantivy_amd_host.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export " + n
    for n in B.EXPORTS:
        assert n in names, n + " bound but not declared in include/"


def test_fails_loudly_without_gpu(B):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = B.lib().tq_init(None, 0, C.byref(ctx))
    assert rc == 5  # TQ_ERR_NO_DEVICE
    assert b"no HIP device" in B.lib().tq_last_error()
    import tantivy_amd

    with pytest.raises(tantivy_amd.TantivyAmdError):
        tantivy_amd.DeviceIndex()


def test_host_bm25_matches_oracle_bitwise(B):
    rng = np.random.default_rng(0)
    for _ in range(50):
        nd = int(rng.integers(10, 50_000_000))
        df = int(rng.integers(1, nd))
        nt = int(nd * rng.uniform(3, 80))
        w, cache = B.bm25_for_terms([df], nd, nt)
        avg = float(np.float32(nt) / np.float32(nd))
        ow = O.bm25_for_one_term(df, nd, avg)
        assert np.float32(w) == np.float32(ow.weight)
        assert np.array_equal(cache, np.array(list(ow.cache), np.float32))
        dfs = [int(x) for x in rng.integers(1, nd, size=3)]
        w3, cache3 = B.bm25_for_terms(dfs, nd, nt, boost=2.0)
        o3 = O.bm25_for_terms(dfs, nd, avg)
        O.lib().to_bm25_boost_by(C.byref(o3), C.c_float(2.0))
        assert np.float32(w3) == np.float32(o3.weight)
        assert np.array_equal(cache3, np.array(list(o3.cache), np.float32))


def test_host_merge_topk_matches_oracle(B):
    rng = np.random.default_rng(1)
    L = B.lib()
    for _ in range(30):
        S, Q, K = int(rng.integers(1, 9)), int(rng.integers(1, 20)), int(rng.integers(1, 12))
        scores = -np.sort(-np.round(rng.random((S, Q, K)), 1).astype(np.float32), axis=2)
        docs = rng.integers(0, 500, size=(S, Q, K)).astype(np.uint32)
        counts = rng.integers(0, K + 1, size=(S, Q)).astype(np.uint32)
        offset, limit = int(rng.integers(0, 5)), int(rng.integers(1, 15))
        hs = np.zeros((Q, limit), np.float32)
        ho = np.zeros((Q, limit), np.uint32)
        hd = np.zeros((Q, limit), np.uint32)
        hc = np.zeros(Q, np.uint32)
        assert L.tq_merge_topk(B._f32(scores), B._u32(docs), B._u32(counts), S, Q, K, offset, limit,
                               B._f32(hs), B._u32(ho), B._u32(hd), B._u32(hc)) == 0
        for q in range(Q):
            hits = [(float(scores[s, q, i]), s, int(docs[s, q, i]))
                    for s in range(S) for i in range(int(counts[s, q]))]
            want = O.merge_top_k(hits, offset, limit)
            got = [(float(hs[q, i]), int(ho[q, i]), int(hd[q, i])) for i in range(int(hc[q]))]
            assert got == want
            assert np.all(hd[q, int(hc[q]):] == 0x7FFFFFFF)


def test_null_arguments_are_errors_not_crashes(B):
    L = B.lib()
    assert L.tq_search_batch(None, None, 1, 10, None, None, None) != 0
    assert L.tq_term_prepare(None, 0, 0, 0, 0, 1, None) != 0
    assert L.tq_last_batch_stats(None, None) != 0
    assert L.tq_set_option(None, b"timing", 1) != 0
    assert L.tq_last_error()
