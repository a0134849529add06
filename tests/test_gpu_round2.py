"""GPU tests closing the holes the round-1 review named: bytes written by released tantivy versions
searched on the device, segments without fieldnorms, legacy (non-strict-delta) blocks, the stream
contract of consecutive batches, two segments searched from two host threads, per-call options."""
import json
import os
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import legacy_posting_list, random_postings

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compat_index.json")


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    from tests.helpers import exhaustive_by_default

    return exhaustive_by_default(tantivy_amd)


def _hits(scores, docs, counts, i):
    return [(float(scores[i, j]), int(docs[i, j])) for j in range(int(counts[i]))]


# ------------------------------------------------------------------ real tantivy bytes
@pytest.mark.parametrize("ver", ["index_v6", "index_v7"])
def test_compat_fixture_searched_on_device(ta, ver):
    """compat_tests.rs:39-57 opens tests/compat_tests_data/index_v6|v7 (written by released
    tantivy versions) and finds the one document with the term query `dateformat`
    (TopDocs::with_limit(1)).  Here the fixture's field sub-files go through tq_segment_upload
    byte for byte, the segment is opened by its TermInfoStore (term ordinal 0) and the same
    query runs on the GPU; the score is the oracle's on the same bytes."""
    with open(GOLD) as f:
        j = json.load(f)[ver]
    files = {k: bytes.fromhex(v) for k, v in j["files"].items()}
    term = O.composite_fields(O.strip_footer(files["term"])[0])
    idx = O.composite_fields(O.strip_footer(files["idx"])[0])
    pos = O.composite_fields(O.strip_footer(files["pos"])[0])
    fnorm = O.composite_fields(O.strip_footer(files["fieldnorm"])[0])
    for key, record in (((0, 0), O.WITH_FREQS_AND_POSITIONS), ((1, 0), O.BASIC)):
        _, store = O.term_dictionary_parts(term[key])
        df, ps, pe, qs, qe = O.term_info_store_get(store, 0)
        body = idx[key]
        fn = np.frombuffer(fnorm[key], np.uint8) if key in fnorm else None
        seg = O.Segment(j["max_doc"], record, np.frombuffer(body, np.uint8),
                        np.frombuffer(pos.get(key, b""), np.uint8), fn,
                        [O.TermInfo(df, ps, pe, qs, qe)], int.from_bytes(body[:8], "little"))
        dev = ta.DeviceIndex([])
        try:
            dev.add_segment(seg, 0, term_info_store=store)  # term id = term ordinal
            for exhaustive in (1, 0):
                dev.set_option("exhaustive", exhaustive)
                sc, ords, docs, cnt = dev.search([(ta.MODE_TERM, [0])], 1)
                want = O.search(seg, [0], O.MODE_OR, 1, pruned=False)
                assert int(cnt[0]) == 1 and _hits(sc, docs, cnt, 0) == want
                assert int(docs[0, 0]) == 0 and int(ords[0, 0]) == 0
                if record != O.BASIC:  # a Must clause over the same term: block_wand_intersection
                    sc2, _, docs2, cnt2 = dev.search([(ta.MODE_AND, [0, 0])], 1)
                    assert _hits(sc2, docs2, cnt2, 0) == O.search(seg, [0, 0], O.MODE_AND, 1, pruned=False)
            gd, gt = dev.decode_postings(0, df)
            assert gd.tolist() == [0] and gt.tolist() == [1]
            if record == O.WITH_FREQS_AND_POSITIONS:
                deltas, n = dev.decode_position_deltas(0, 4)
                assert n == 1 and deltas.tolist() == [0]
        finally:
            dev.close()


# ------------------------------------------------------------------ fieldnorm == NULL
def test_search_without_fieldnorms(ta):
    """FieldNormReader::constant(max_doc, 1) (term_weight.rs:209-219): every kernel takes the
    constant-id branch of fieldnorm_id()."""
    rng = np.random.default_rng(77)
    md = 60_000
    dfs = [30_000, 9_000, 2_500, 700, 129, 128, 5, 20_000]
    lists = [random_postings(rng, md, df, max_tf=6) for df in dfs]
    positions = [[sorted(rng.choice(40, size=tf, replace=False).tolist()) for _, tf in pl] for pl in lists]
    seg = O.build_segment(md, lists, None, record_option=O.WITH_FREQS_AND_POSITIONS,
                          positions=positions, total_num_tokens=md * 7, avg_fieldnorm=7.0)
    assert seg.fieldnorm is None
    dev = ta.DeviceIndex([seg])
    try:
        qs = [(O.MODE_AND, [0, 1]), (O.MODE_AND, [7, 0]), (O.MODE_AND, [2, 3]), (O.MODE_AND, [0, 1, 7]),
              (O.MODE_OR, [0, 3]), (O.MODE_OR, [1, 2, 3, 4, 6]), (O.MODE_OR, [5]),
              (O.MODE_PHRASE, [0, 7]), (O.MODE_PHRASE, [1, 0, 7])]
        for exhaustive in (1, 0):
            for use_dense in (1, 0):
                dev.set_option("exhaustive", exhaustive)
                dev.set_option("use_dense", use_dense)
                for k in (1, 10, 100):
                    sc, _, docs, cnt = dev.search(qs, k)
                    for i, q in enumerate(qs):
                        want = O.search(seg, q[1], q[0], k, pruned=False)
                        got = _hits(sc, docs, cnt, i)
                        assert [d for _, d in got] == [d for _, d in want], (q, k, exhaustive)
                        for (gs, _), (ws, _) in zip(got, want):
                            if q[0] == O.MODE_OR and len(q[1]) > 2:
                                assert abs(gs - ws) <= 1e-5 * abs(ws)
                            else:
                                assert np.float32(gs) == np.float32(ws), (q, gs, ws)
    finally:
        dev.close()


# ------------------------------------------------------------------ legacy blocks (flag 0)
def test_legacy_non_strict_blocks_on_device(ta):
    """Blocks written before strict deltas (width byte without bit 6: D1 deltas seeded with the
    previous block's last doc, raw tfs) — compression/mod.rs:124-125.  The oracle's serializer
    never writes them, so the list is packed by hand (tests/helpers.py)."""
    rng = np.random.default_rng(31)
    md = 200_000
    old = random_postings(rng, md, 128 * 9 + 57, max_tf=9)
    old[0] = (0, 3)                      # doc 0 first: delta 0 from the seed 0 is legal only here
    old = sorted(dict(old).items())
    new = random_postings(rng, md, 40_000, max_tf=5)
    ref = O.build_segment(md, [old, new], rng.integers(1, 300, size=md).tolist())
    legacy = legacy_posting_list(old)
    t_new = ref.terms[1]
    new_bytes = bytes(ref.idx[8 + t_new.postings_start: 8 + t_new.postings_end])
    body = bytes(ref.idx[:8]) + legacy + new_bytes
    terms = [O.TermInfo(len(old), 0, len(legacy), 0, 0),
             O.TermInfo(len(new), len(legacy), len(legacy) + len(new_bytes), 0, 0)]
    seg = O.Segment(md, O.WITH_FREQS, np.frombuffer(body, np.uint8), np.zeros(0, np.uint8),
                    ref.fieldnorm, terms, ref.total_num_tokens)
    od, ot = O.decode_postings(seg, 0)      # the oracle reads the flag (to_postings.c)
    assert od.tolist() == [d for d, _ in old] and ot.tolist() == [t for _, t in old]
    dev = ta.DeviceIndex([seg])
    try:
        for use_dpp in (1, 0):
            dev.set_option("use_dpp", use_dpp)
            gd, gt = dev.decode_postings(0, len(old))
            assert gd.tolist() == [d for d, _ in old] and gt.tolist() == [t for _, t in old]
        dev.set_option("use_dpp", 1)
        qs = [(O.MODE_AND, [0, 1]), (O.MODE_OR, [0, 1]), (O.MODE_OR, [0])]
        for exhaustive in (1, 0):
            for use_dense in (1, 0):  # legacy list as leader (stage A) and as probed list (find_in_blocks)
                dev.set_option("exhaustive", exhaustive)
                dev.set_option("use_dense", use_dense)
                sc, _, docs, cnt = dev.search(qs, 20)
                for i, q in enumerate(qs):
                    # block-max bytes are 0 (= unknown) in these entries: scores must still be exact
                    assert _hits(sc, docs, cnt, i) == O.search(ref, q[1], q[0], 20, pruned=False)
    finally:
        dev.close()


# ------------------------------------------------------------------ streams and threads
def test_consecutive_batches_on_different_streams(ta):
    """The per-segment scratch is shared by consecutive batches: a batch enqueued on another
    stream must wait for the previous one (tq_api.cpp order_after_last_batch)."""
    import torch

    seg = O.synth_segment(2_000_000, n_terms=64)
    dev = ta.DeviceIndex([seg])
    try:
        qa = [(O.MODE_AND, q.tolist()) for q in O.zipf_queries(3000, 2, 64, seed=1)]
        qb = [(O.MODE_OR, q.tolist()) for q in O.zipf_queries(300, 3, 64, seed=2)]
        k = 10
        want = {}
        for name, qs in (("a", qa), ("b", qb)):
            sc, _, dc, ct = dev.search(qs, k)
            want[name] = (sc.copy(), dc.copy(), ct.copy())
        streams = [torch.cuda.Stream(), torch.cuda.Stream(), None]
        outs = []
        for it in range(6):
            name, qs = ("a", qa) if it % 2 == 0 else ("b", qb)
            st = streams[it % 3]
            d_sc = torch.empty((len(qs), k), dtype=torch.float32, device="cuda")
            d_dc = torch.empty((len(qs), k), dtype=torch.int32, device="cuda")
            d_ct = torch.empty(len(qs), dtype=torch.int32, device="cuda")
            dev.prepare(qs)
            dev.collect_segment_prepared_device(0, k, d_sc, d_dc, d_ct,
                                                st.cuda_stream if st is not None else None)
            outs.append((name, d_sc, d_dc, d_ct))  # no synchronisation between the batches
        dev.last_batch_stats()  # waits for whatever the segment has in flight, on any stream
        torch.cuda.synchronize()
        for name, d_sc, d_dc, d_ct in outs:
            sc, dc, ct = want[name]
            assert np.array_equal(d_ct.cpu().numpy().view(np.uint32), ct)
            assert np.array_equal(d_dc.cpu().numpy().view(np.uint32), dc)
            assert np.array_equal(d_sc.cpu().numpy(), sc)
    finally:
        dev.close()


def test_two_segments_from_two_threads(ta):
    """include/tantivy_amd.h: different segments may be searched concurrently (tantivy's one
    task per segment, executor.rs:61-104)."""
    segs = [O.synth_segment(1_500_000, n_terms=48, segment_ord=o) for o in range(2)]
    devs = [ta.DeviceIndex([s]) for s in segs]
    try:
        qs = [(O.MODE_AND, q.tolist()) for q in O.zipf_queries(1500, 2, 48, seed=9)]
        qs += [(O.MODE_OR, q.tolist()) for q in O.zipf_queries(200, 4, 48, seed=10)]
        want = [d.search(qs, 10) for d in devs]
        got = [[None] * 8, [None] * 8]
        errs = []

        def run(i):
            try:
                for it in range(8):
                    got[i][it] = devs[i].search(qs, 10)
            except Exception as e:  # pragma: no cover
                errs.append(e)

        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(2):
            for it in range(8):
                for a, b in zip(got[i][it], want[i]):
                    assert np.array_equal(a, b)
        # spot check against the oracle
        for i in range(2):
            sc, _, dc, ct = want[i]
            for qi in (0, 500, 1499, 1600):
                w = O.search(segs[i], qs[qi][1], qs[qi][0], 10, pruned=False)
                g = _hits(sc, dc, ct, qi)
                assert [d for _, d in g] == [d for _, d in w]
    finally:
        for d in devs:
            d.close()


# ------------------------------------------------------------------ per-call options, argument checks
def test_per_call_options_and_argument_checks(ta):
    import ctypes as C

    from tantivy_amd import binding as B

    seg = O.synth_segment(300_000, n_terms=32)
    dev = ta.DeviceIndex([seg])
    try:
        assert ta.MODE_BOOL == O.MODE_BOOL == 3 and ta.MODE_TERM == 4
        with pytest.raises(ta.TantivyAmdError):
            dev.prepare([(7, [0, 1])])  # unknown mode
        qs = [(O.MODE_AND, [0, 1]), (O.MODE_AND, [2, 9])]
        ws = [[float(w.weight) for w in O.default_weights(seg, q[1], q[0])] for q in qs]
        _, cache = ta.bm25_for_terms([seg.terms[0].doc_freq], seg.max_doc, seg.total_num_tokens)
        base = dev.raw_search(qs, ws, cache, 10)
        # exhaustive / pruned per call give the same top-k and leave the segment option alone
        for exh in (0, 1, -1):
            got = dev.raw_search(qs, ws, cache, 10, opts=(exh, 0 if exh == 0 else B.OPT_DEFAULT))
            for a, b in zip(got, base):
                assert np.array_equal(a, b)
        counts = dev.raw_count(qs, ws, cache)   # exhaustive for this call only
        n0 = len(O.match_all(seg, [0, 1], O.MODE_AND)[0])
        assert int(counts[0]) == n0
        with pytest.raises(ta.TantivyAmdError):
            dev.raw_search(qs, ws, cache, 10, opts=(5, 0))
        with pytest.raises(ta.TantivyAmdError):   # NaN / inf weights are rejected
            dev.raw_search(qs, [[float("nan"), 1.0], ws[1]], cache, 10)
        with pytest.raises(ta.TantivyAmdError):
            dev.raw_search(qs, [[float("inf"), 1.0], ws[1]], cache, 10)
        # BoostQuery around a PhraseQuery scales its score
        segp = O.synth_segment(100_000, n_terms=16, with_positions=True, phrase_terms=8)
        devp = ta.DeviceIndex([segp])
        try:
            s1, _, d1, c1 = devp.search([(O.MODE_PHRASE, [0, 1])], 5)
            s2, _, d2, c2 = devp.search([(O.MODE_PHRASE, [0, 1], None, {"boosts": [2.5]})], 5)
            assert int(c1[0]) > 0 and np.array_equal(d1, d2)
            w = np.float32(s1[0, 0])
            assert np.all(np.abs(s2[0, :int(c2[0])] - s1[0, :int(c1[0])] * np.float32(2.5)) <=
                          1e-6 * np.abs(s2[0, :int(c2[0])])) and w > 0
        finally:
            devp.close()
    finally:
        dev.close()


# ------------------------------------------------------------------ RCCL behind the C ABI
def test_rccl_allgather_through_the_c_abi(ta):
    """tq_comm_unique_id / tq_comm_init / tq_allgather_topk on a one-rank communicator (the GPU
    box has one GPU): the gathered slabs equal the inputs, and a ShardRunner that is forced
    through the exchange (collect_segment x 2 local segments -> all-gather -> merge_top_k, one
    stream) returns what Searcher::search over the same two segments returns."""
    import torch

    from tantivy_amd import distributed as D

    segs = [O.synth_segment(400_000, n_terms=32, segment_ord=o) for o in range(2)]
    ref = ta.DeviceIndex(segs, devices=[0])
    qs = [(O.MODE_AND, q.tolist()) for q in O.zipf_queries(500, 2, 32, seed=3)]
    qs += [(O.MODE_OR, q.tolist()) for q in O.zipf_queries(100, 3, 32, seed=4)]
    want = ref.search(qs, 10)
    ref.close()
    run = D.ShardRunner(segs, 0, 0, 1, force_exchange=True)
    comm = D.Comm(run.dev.ctx, 0, 0, 1, lambda raw: raw)
    try:
        assert comm.library
        run.comm = comm
        run.prepare(qs, 10)
        for _ in range(3):
            run.enqueue()
        run.synchronize()
        got = run.results()
        for a, b in zip(got, want):
            assert np.array_equal(a.view(np.uint32) if a.dtype != np.float32 else a,
                                  b.view(np.uint32) if b.dtype != np.float32 else b)
        # raw call: [rows][k] slabs in, [1][rows][k] out
        sc = torch.rand((7, 5), device="cuda")
        dc = torch.randint(0, 1000, (7, 5), dtype=torch.int32, device="cuda")
        ct = torch.randint(0, 6, (7,), dtype=torch.int32, device="cuda")
        out = (torch.zeros((1, 7, 5), device="cuda"), torch.zeros((1, 7, 5), dtype=torch.int32, device="cuda"),
               torch.zeros((1, 7), dtype=torch.int32, device="cuda"))
        comm.allgather_topk(sc, dc, ct, out, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(out[0][0], sc) and torch.equal(out[1][0], dc) and torch.equal(out[2][0], ct)
    finally:
        comm.close()
        run.close()


# ------------------------------------------------------------------ tq_term_prepare on the device
def _search_all(dev, qs, k):
    sc, _, dc, ct = dev.search(qs, k)
    return [(sc[i, :int(ct[i])].tolist(), dc[i, :int(ct[i])].tolist()) for i in range(len(qs))]


def test_device_side_term_prepare_equals_host_walk(ta):
    """option "device_prepare": the skip lists / positions headers are walked by tq_prepare.hip in
    HBM and the dense-list tables built by device scans; everything a search or a decode returns
    must equal the host walk's (and the oracle's), list shapes of the edge-case test included."""
    rng = np.random.default_rng(123)
    md = 300_000
    dfs = [100, 127, 128, 129, 255, 256, 257, 1, 5000, 40_000, 150_000, 64 * 128, 64 * 128 + 1, 65 * 128 + 77]
    lists = [random_postings(rng, md, df, max_tf=7) for df in dfs]
    lists[7] = [(md - 1, 3)]
    positions = [[sorted(rng.choice(60, size=tf, replace=False).tolist()) for _, tf in pl] for pl in lists]
    fieldnorms = rng.integers(1, 500, size=md).tolist()
    for record, pos in ((O.WITH_FREQS_AND_POSITIONS, positions), (O.WITH_FREQS, None), (O.BASIC, None)):
        pls = lists if record != O.BASIC else [[(d, 1) for d, _ in pl] for pl in lists]
        seg = O.build_segment(md, pls, fieldnorms, record_option=record, positions=pos)
        qs = [(O.MODE_AND, [9, 10]), (O.MODE_AND, [8, 10]), (O.MODE_AND, [4, 5, 10]), (O.MODE_OR, [0, 1, 2, 3]),
              (O.MODE_OR, [7, 8, 9]), (O.MODE_AND, [11, 12]), (O.MODE_OR, [13, 6]), (O.MODE_AND, [13, 9])]
        if record == O.WITH_FREQS_AND_POSITIONS:
            qs += [(O.MODE_PHRASE, [9, 10]), (O.MODE_PHRASE, [10, 9, 8]), (O.MODE_PHRASE, [8, 13])]
        results = []
        for device_prepare in (0, 1):
            dev = ta.DeviceIndex([seg])
            try:
                dev.set_option("device_prepare", device_prepare)
                got = {ex: _search_all(dev, qs, 10) for ex in (1, 0) if not dev.set_option("exhaustive", ex)}
                dec = [tuple(a.tolist() for a in dev.decode_postings(t, len(pls[t]))) for t in range(len(pls))]
                pd = None
                if record == O.WITH_FREQS_AND_POSITIONS:
                    pd = [dev.decode_position_deltas(t, sum(tf for _, tf in pls[t]))[0].tolist()
                          for t in range(len(pls))]
                results.append((got, dec, pd))
            finally:
                dev.close()
        assert results[0] == results[1]
        for i, q in enumerate(qs):  # and the oracle
            want = O.search(seg, q[1], q[0], 10, pruned=False)
            assert results[1][0][1][i][1] == [d for _, d in want], q
        for t, pl in enumerate(pls):
            assert results[1][1][t][0] == [d for d, _ in pl]


def test_device_side_prepare_reports_corrupt_lists(ta):
    """the device walk maps malformed bytes to TQ_ERR_FORMAT like the host walk"""
    rng = np.random.default_rng(5)
    seg = O.build_segment(50_000, [random_postings(rng, 50_000, 3000, max_tf=5)], rng.integers(1, 50, size=50_000).tolist())
    bad = O.Segment(seg.max_doc, seg.record_option, seg.idx[: seg.idx_len].copy(), np.zeros(0, np.uint8),
                    seg.fieldnorm, seg.terms, seg.total_num_tokens)
    bad.idx[8 + 3 + 8 * 5:8 + 3 + 8 * 5 + 4] = 0  # last_doc of skip entry 5 := 0 (not increasing)
    for device_prepare in (0, 1):
        dev = ta.DeviceIndex([bad])
        try:
            dev.set_option("device_prepare", device_prepare)
            with pytest.raises(ta.TantivyAmdError):
                dev.search([(O.MODE_OR, [0])], 5)
            # a caller that retries the failing term does not grow the segment's tables (the failed
            # preparation's blob is handed back)
            before = dev.segment_stats(0)["term_table_bytes"]
            for _ in range(20):
                with pytest.raises(ta.TantivyAmdError):
                    dev.search([(O.MODE_OR, [0])], 5)
            assert dev.segment_stats(0)["term_table_bytes"] == before
        finally:
            dev.close()


def test_segment_encoded_and_opened_on_the_device(ta):
    """Segment finalisation with the index bytes never leaving the GPU: postings + positions
    encoded by tq_encode_*_device into device buffers, adopted with tq_segment_upload_device
    (device-to-device), terms prepared by the device walk; only TermInfo ranges (the term
    dictionary's values) and 48-byte fact records cross to the host.  Answers like the oracle."""
    import ctypes as C

    import torch

    from tantivy_amd import binding as B

    seg = O.synth_segment(120_000, n_terms=24, with_positions=True, phrase_terms=8)
    starts, docs, tfs, pstarts, deltas = [0], [], [], [0], []
    for t in range(len(seg.terms)):
        d, f = O.decode_postings(seg, t)
        docs.append(d)
        tfs.append(f)
        starts.append(starts[-1] + len(d))
        ps, _ = O.decode_positions(seg, t, int(f.sum()))
        dl = np.diff(ps.astype(np.int64), prepend=0)
        first = np.cumsum(f.astype(np.int64)) - f
        dl[first] = ps[first]
        deltas.append(dl.astype(np.uint32))
        pstarts.append(pstarts[-1] + len(dl))
    n = len(seg.terms)
    avg = float(np.float32(seg.total_num_tokens) / np.float32(seg.max_doc))
    L = B.lib()
    dev = ta.DeviceIndex([])
    enc = ta.Encoder(0)
    try:
        cuda = torch.device("cuda", 0)
        h_starts = np.array(starts, np.uint64)
        h_pstarts = np.array(pstarts, np.uint64)
        d_starts = torch.from_numpy(h_starts.view(np.int64)).to(cuda)
        d_pstarts = torch.from_numpy(h_pstarts.view(np.int64)).to(cuda)
        d_docs = torch.from_numpy(np.concatenate(docs).view(np.int32)).to(cuda)
        d_tfs = torch.from_numpy(np.concatenate(tfs).view(np.int32)).to(cuda)
        d_deltas = torch.from_numpy(np.concatenate(deltas).view(np.int32)).to(cuda)
        d_fn = torch.from_numpy(seg.fieldnorm).to(cuda)
        cap = 8 * int(starts[-1]) + 4096
        d_idx = torch.zeros(8 + cap, dtype=torch.uint8, device=cuda)
        d_idx[:8] = torch.from_numpy(np.frombuffer(int(seg.total_num_tokens).to_bytes(8, "little"), np.uint8).copy()).to(cuda)
        d_ots = torch.zeros(n + 1, dtype=torch.int64, device=cuda)
        out_len = C.c_uint64()
        B._check(L.tq_encode_postings_device(enc.raw, n, h_starts.ctypes.data, d_starts.data_ptr(),
                                             d_docs.data_ptr(), d_tfs.data_ptr(), d_fn.data_ptr(),
                                             seg.max_doc, C.c_float(avg), O.WITH_FREQS_AND_POSITIONS,
                                             d_idx.data_ptr() + 8, cap, d_ots.data_ptr(),
                                             C.byref(out_len), None))
        idx_len = 8 + out_len.value
        pcap = 8 * int(pstarts[-1]) + 4096
        d_pos = torch.zeros(pcap, dtype=torch.uint8, device=cuda)
        d_pts = torch.zeros(n + 1, dtype=torch.int64, device=cuda)
        plen = C.c_uint64()
        B._check(L.tq_encode_positions_device(enc.raw, n, h_pstarts.ctypes.data, d_pstarts.data_ptr(),
                                              d_deltas.data_ptr(), d_pos.data_ptr(), pcap,
                                              d_pts.data_ptr(), C.byref(plen), None))
        torch.cuda.synchronize()
        # the device bytes equal the oracle's serializer output (checked once; not needed to search)
        assert bytes(d_idx[:idx_len].cpu().numpy()) == bytes(seg.idx[: seg.idx_len])
        assert bytes(d_pos[:plen.value].cpu().numpy()) == bytes(seg.pos[: seg.pos_len])
        ots = d_ots.cpu().numpy().view(np.uint64)   # TermInfo ranges: the term dictionary's values
        pts = d_pts.cpu().numpy().view(np.uint64)
        infos = [(len(docs[t]), int(ots[t]), int(ots[t + 1]), int(pts[t]), int(pts[t + 1])) for t in range(n)]
        store = ta.TermInfoStore.serialize(infos)
        dev.add_segment_device(seg.max_doc, O.WITH_FREQS_AND_POSITIONS, d_idx[:idx_len], d_pos[:plen.value],
                               d_fn, seg.total_num_tokens, store)
        queries = [(O.MODE_AND, [0, 1]), (O.MODE_AND, [3, 7, 11]), (O.MODE_OR, [2, 5, 19]),
                   (O.MODE_PHRASE, [0, 1, 2]), (O.MODE_OR, [17]), (O.MODE_PHRASE, [6, 7]), (O.MODE_AND, [20, 23])]
        for exhaustive in (1, 0):
            dev.set_option("exhaustive", exhaustive)
            scores, _, dd, counts = dev.search(queries, 10)
            for i, (mode, terms) in enumerate(queries):
                want = O.search(seg, terms, mode, 10, pruned=False)
                got = _hits(scores, dd, counts, i)
                assert [d for _, d in got] == [d for _, d in want], (mode, terms)
                for (gs, _), (ws, _) in zip(got, want):
                    assert abs(gs - ws) <= 1e-5 * abs(ws)
    finally:
        enc.close()
        dev.close()


# ------------------------------------------------------------------ union kernel: bitmap sweep
@pytest.mark.parametrize("k", [17, 64, 100, 128])
def test_union_bitmap_sweep_matches_exhaustive_and_oracle(ta, k):
    """Pure unions whose leaders are dense enough to take the bitmap sweep of tq_union.hip (a
    leader with a bitmap and 1..5 dense lists after it, k > 16): pruned == exhaustive bit for bit,
    with and without deletes, and equal to the oracle's union (block_wand_union.rs semantics;
    3+ term sums within 1e-5).  The sparse terms 40.. make tiles of both kinds meet in one query."""
    from tests.test_gpu_parity import _assert_hits_close, _alive_bytes

    seg = O.synth_segment(300_000, n_terms=64)
    rng = np.random.default_rng(1000 + k)
    qs = [[0, 1, 2, 3, 4], [1, 2, 3, 5, 8], [0, 2, 4, 6, 7, 9], [3, 5, 7, 40, 50], [2, 6, 45, 55, 63],
          [0, 1], [4, 9, 60]]
    qs += [sorted(rng.choice(12, size=5, replace=False).tolist()) for _ in range(6)]
    batch = [(O.MODE_OR, q) for q in qs]
    dev = ta.DeviceIndex([seg])
    try:
        for deleted in (None, set(rng.choice(seg.max_doc, size=seg.max_doc // 4, replace=False).tolist())):
            if deleted is not None:
                dev.set_alive_bitset(_alive_bytes(seg.max_doc, deleted))
            dev.set_option("exhaustive", 1)
            s, _, d, c = dev.search(batch, k)
            full = [_hits(s, d, c, i) for i in range(len(qs))]
            dev.set_option("exhaustive", 0)
            s, _, d, c = dev.search(batch, k)
            pruned = [_hits(s, d, c, i) for i in range(len(qs))]
            assert pruned == full
            for q, got in zip(qs, full):
                docs, scores = O.match_all(seg, q, O.MODE_OR)
                hits = [(float(sc), int(doc)) for doc, sc in zip(docs.tolist(), scores.tolist())
                        if deleted is None or doc not in deleted]
                hits.sort(key=lambda h: (-h[0], h[1]))
                _assert_hits_close(got, hits[:k])
    finally:
        dev.close()


# ------------------------------------------------------------------ nested boolean queries that flatten
def test_nested_must_queries_are_hoisted(ta):
    """`+a +(+b +c)`, `+a +(+b -c)`, `+a +(+b c)`, `+(+a +b) +(c OR d)`: a Must clause holding a
    BooleanQuery with a Must term is an Intersection with that query's scorer
    (boolean_weight.rs:308-431) — the docs and score terms of the flat query with the nested
    clauses hoisted; only the association of the f32 sum differs (1e-5).  An intersection inside a
    union / under MustNot stays Unsupported (the caller keeps tantivy's CPU scorer)."""
    from tests.test_gpu_parity import _assert_hits_close

    M, S, N = ta.MUST, ta.SHOULD, ta.MUST_NOT
    seg = O.synth_segment(150_000, n_terms=40)
    nested = [
        (ta.MODE_BOOL, [3, 5, 9], [M, M, M], [0, 1, 1], 0, {"nested_occurs": [255, 1, 1]}),
        (ta.MODE_BOOL, [2, 6, 1], [M, M, M], [0, 1, 1], 0, {"nested_occurs": [255, 1, 2]}),
        (ta.MODE_BOOL, [4, 7, 0], [M, M, M], [0, 1, 1], 0, {"nested_occurs": [255, 1, 0]}),
        (ta.MODE_BOOL, [8, 3, 20, 30], [M, M, M, M], [0, 0, 1, 1], 0, {"nested_occurs": [1, 1, 255, 255]}),
        (ta.MODE_BOOL, [1, 10, 12, 2], [M, M, M, S], [0, 1, 1, 2], 0, {"nested_occurs": [255, 1, 1, 255]}),
    ]
    flat = [
        (ta.MODE_BOOL, [3, 5, 9], [M, M, M], None, 0),
        (ta.MODE_BOOL, [2, 6, 1], [M, M, N], None, 0),
        (ta.MODE_BOOL, [4, 7, 0], [M, M, S], None, 0),
        (ta.MODE_BOOL, [8, 3, 20, 30], [M, M, M, M], [0, 1, 2, 2], 0),
        (ta.MODE_BOOL, [1, 10, 12, 2], [M, M, M, S], None, 0),
    ]
    dev = ta.DeviceIndex([seg])
    try:
        for k in (10, 100):
            for ex in (1, 0):
                dev.set_option("exhaustive", ex)
                s, _, d, c = dev.search(nested, k)
                got = [_hits(s, d, c, i) for i in range(len(nested))]
                s, _, d, c = dev.search(flat, k)
                want = [_hits(s, d, c, i) for i in range(len(flat))]
                for g, w, q in zip(got, want, nested):
                    assert len(g) > 0, q
                    _assert_hits_close(g, w)
        # against the oracle's scorer tree on the flat forms
        dev.set_option("exhaustive", 1)
        s, _, d, c = dev.search(nested, 10)
        for i, q in enumerate(flat):
            want = O.bool_search(seg, q[1], q[2], 10, q[3], q[4])
            _assert_hits_close(_hits(s, d, c, i), want)
        # not hoistable — an intersection inside a union, under MustNot, or with a minimum on the parent: since
        # round 5 these run on the device over the lists' bitmaps (tq_tree.hip; tests/test_gpu_tree.py); doc ids as the
        # nested-tree oracle's
        trees = [((ta.MODE_BOOL, [1, 2, 3], [S, S, S], [0, 1, 1], 0, {"nested_occurs": [255, 1, 1]}),
                  [(S, 1), (S, [(M, 2), (M, 3)], 0)], 0),
                 ((ta.MODE_BOOL, [1, 2, 3], [M, N, N], [0, 1, 1], 0, {"nested_occurs": [255, 1, 1]}),
                  [(M, 1), (N, [(M, 2), (M, 3)], 0)], 0),
                 ((ta.MODE_BOOL, [1, 2, 3, 4], [M, M, M, S], [0, 1, 1, 2], 1, {"nested_occurs": [255, 1, 1, 255]}),
                  [(M, 1), (M, [(M, 2), (M, 3)], 0), (S, 4)], 1)]
        for q, tree, msm in trees:
            s, _, d, c = dev.search([q], 10)
            want = O.tree_search(seg, tree, 10, msm)
            assert sorted(int(d[0, j]) for j in range(int(c[0]))) == sorted(x for _, x in want), q
    finally:
        dev.close()
