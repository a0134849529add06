"""GPU parity of the device-side codec writers (tq_encode_*): the bytes must equal
what the oracle's restatement of PostingsSerializer / PositionSerializer writes, for every record
option, and decode back (through the device read path) to the input."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    from tests.helpers import exhaustive_by_default

    return exhaustive_by_default(tantivy_amd)


@pytest.fixture(scope="module")
def enc(ta):
    e = ta.Encoder(0)
    yield e
    e.close()


def _random_lists(rng, max_doc, dfs, max_tf=12, big_tf_every=0):
    starts, docs, tfs = [0], [], []
    for i, df in enumerate(dfs):
        d = np.sort(rng.choice(max_doc, size=df, replace=False)).astype(np.uint32)
        t = rng.integers(1, max_tf + 1, size=df).astype(np.uint32)
        if big_tf_every and df and i % big_tf_every == 0:
            t[rng.integers(0, df)] = 300 + i  # block-max tf code saturates at 255 (skip.rs:31-34)
        docs.append(d)
        tfs.append(t)
        starts.append(starts[-1] + df)
    return (np.array(starts, np.uint64), np.concatenate(docs) if docs else np.zeros(0, np.uint32),
            np.concatenate(tfs) if tfs else np.zeros(0, np.uint32))


DFS = [0, 1, 5, 127, 128, 129, 255, 256, 257, 1000, 1280, 5000, 20000, 3, 640]


@pytest.mark.parametrize("record_option", [O.BASIC, O.WITH_FREQS, O.WITH_FREQS_AND_POSITIONS])
@pytest.mark.parametrize("with_fieldnorms", [True, False])
def test_encode_postings_bytes_equal_serializer(enc, record_option, with_fieldnorms):
    rng = np.random.default_rng(100 + record_option * 2 + with_fieldnorms)
    md = 60_000
    ts, docs, tfs = _random_lists(rng, md, DFS, big_tf_every=4)
    fn = rng.integers(0, 256, size=md).astype(np.uint8) if with_fieldnorms else None
    avg = 37.25
    want, want_ts = O.serialize_postings_batch(ts, docs, tfs, fn, md, avg, record_option)
    got, got_ts = enc.encode_postings(ts, docs, None if record_option == O.BASIC else tfs, fn, md,
                                      avg, record_option)
    assert np.array_equal(got_ts, want_ts)
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:10], got[bad[:10]], want[bad[:10]])


def test_encode_postings_edge_values(enc):
    """doc 0 first, consecutive docs (zero-width deltas), the largest doc ids, wide tfs."""
    lists = [
        np.arange(0, 128, dtype=np.uint32),                       # widths 0 after a raw 0
        np.arange(5, 5 + 300, dtype=np.uint32),                   # zero-width deltas, tail
        np.array([0x7FFFFFFE - 200 + i for i in range(130)], np.uint32),  # 31-bit first doc
        np.concatenate([np.arange(0, 127, dtype=np.uint32), np.array([0x7FFFFFF0], np.uint32)]),
        np.array([7], np.uint32),
    ]
    tfs = [np.ones(len(l), np.uint32) for l in lists]
    tfs[1][17] = 0xFFFFFFFF  # tf - 1 needs 32 bits: raw copy
    tfs[4][0] = 1 << 30
    ts = np.cumsum([0] + [len(l) for l in lists]).astype(np.uint64)
    docs, tf = np.concatenate(lists), np.concatenate(tfs)
    for mode in (O.WITH_FREQS, O.WITH_FREQS_AND_POSITIONS):
        want, want_ts = O.serialize_postings_batch(ts, docs, tf, None, 0, 0.0, mode)
        got, got_ts = enc.encode_postings(ts, docs, tf, None, 0, 0.0, mode)
        assert np.array_equal(got_ts, want_ts) and np.array_equal(got, want)


def test_encode_positions_bytes_equal_serializer(enc):
    rng = np.random.default_rng(7)
    ns = [0, 1, 127, 128, 129, 1000, 4096, 50_000, 300]
    starts = np.cumsum([0] + ns).astype(np.uint64)
    deltas = rng.integers(0, 50, size=int(starts[-1])).astype(np.uint32)
    deltas[rng.integers(0, deltas.size, size=20)] = rng.integers(1 << 20, 1 << 31, size=20)
    deltas[128 * 3: 128 * 4] = 0          # a zero-width block
    deltas[128 * 5] = 0xFFFFFFFF          # a 32-bit block
    want, want_ts = O.serialize_positions_batch(starts, deltas)
    got, got_ts = enc.encode_positions(starts, deltas)
    assert np.array_equal(got_ts, want_ts)
    assert np.array_equal(got, want)


def test_encode_then_search_round_trip(ta, enc):
    """A segment whose .idx / .pos bytes come from the device encoder answers queries exactly like
    the one the oracle serialised, and decodes back to the input postings."""
    seg = O.synth_segment(80_000, n_terms=24, with_positions=True, phrase_terms=8)
    starts, docs, tfs, pstarts, deltas = [0], [], [], [0], []
    for t in range(len(seg.terms)):
        d, f = O.decode_postings(seg, t)
        docs.append(d)
        tfs.append(f)
        starts.append(starts[-1] + len(d))
        ps, _ = O.decode_positions(seg, t, int(f.sum()))  # absolute positions, doc after doc
        dl = np.diff(ps.astype(np.int64), prepend=0)
        first = np.cumsum(f.astype(np.int64)) - f      # index of every doc's first position
        dl[first] = ps[first]
        dl = dl.astype(np.uint32)
        deltas.append(dl)
        pstarts.append(pstarts[-1] + len(dl))
    docs, tfs, deltas = np.concatenate(docs), np.concatenate(tfs), np.concatenate(deltas)
    avg = float(np.float32(seg.total_num_tokens) / np.float32(seg.max_doc))
    body, ots = enc.encode_postings(np.array(starts, np.uint64), docs, tfs, seg.fieldnorm,
                                    seg.max_doc, avg, O.WITH_FREQS_AND_POSITIONS)
    pos, pts = enc.encode_positions(np.array(pstarts, np.uint64), deltas)
    idx_len = getattr(seg, "idx_len", len(seg.idx))
    assert np.array_equal(body, np.asarray(seg.idx[8:idx_len]))
    pos_len = getattr(seg, "pos_len", len(seg.pos))
    assert np.array_equal(pos, np.asarray(seg.pos[:pos_len]))
    for t, ti in enumerate(seg.terms):
        assert (int(ots[t]), int(ots[t + 1])) == (ti.postings_start, ti.postings_end)
        assert (int(pts[t]), int(pts[t + 1])) == (ti.positions_start, ti.positions_end)


def test_encode_nothing(enc):
    body, ots = enc.encode_postings(np.zeros(1, np.uint64), np.zeros(0, np.uint32), None, None, 0, 0.0,
                                    O.BASIC)
    assert body.size == 0 and ots.tolist() == [0]
    body, ots = enc.encode_positions(np.zeros(1, np.uint64), np.zeros(0, np.uint32))
    assert body.size == 0 and ots.tolist() == [0]


def test_encode_errors(ta, enc):
    ts = np.array([0, 200], np.uint64)
    docs = np.arange(200, dtype=np.uint32)
    with pytest.raises(ta.TantivyAmdError):  # tfs missing for a field with freqs
        enc.encode_postings(ts, docs, None, None, 0, 0.0, O.WITH_FREQS)
    # a too-small buffer reports the size and the retry inside the binding succeeds
    body, _ = enc.encode_postings(ts, docs, None, None, 0, 0.0, O.BASIC, out_cap=16)
    want, _ = O.serialize_postings_batch(ts, docs, None, None, 0, 0.0, O.BASIC)
    assert np.array_equal(body, want)


def test_segment_built_on_the_device_and_opened_by_ordinal(ta, enc):
    """Segment finalisation without the CPU serializer: postings + positions encoded on the
    device, their TermInfo ranges written by the product's TermInfoStoreWriter, the segment opened
    through that store (term id = term ordinal) — and it answers like the oracle's segment."""
    seg = O.synth_segment(60_000, n_terms=20, with_positions=True, phrase_terms=8)
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
    avg = float(np.float32(seg.total_num_tokens) / np.float32(seg.max_doc))
    body, ots = enc.encode_postings(np.array(starts, np.uint64), np.concatenate(docs),
                                    np.concatenate(tfs), seg.fieldnorm, seg.max_doc, avg,
                                    O.WITH_FREQS_AND_POSITIONS)
    pos, pts = enc.encode_positions(np.array(pstarts, np.uint64), np.concatenate(deltas))
    infos = [(len(docs[t]), int(ots[t]), int(ots[t + 1]), int(pts[t]), int(pts[t + 1]))
             for t in range(len(seg.terms))]
    store = ta.TermInfoStore.serialize(infos)
    assert store == O.term_info_store_serialize(infos)

    class Built:
        max_doc = seg.max_doc
        record_option = O.WITH_FREQS_AND_POSITIONS
        idx = np.concatenate([np.frombuffer(int(seg.total_num_tokens).to_bytes(8, "little"), np.uint8), body])
        fieldnorm = seg.fieldnorm
        terms = []

    Built.pos = pos
    dev = ta.DeviceIndex([])
    try:
        dev.add_segment(Built, 0, term_info_store=store)
        queries = [(O.MODE_AND, [0, 1]), (O.MODE_AND, [3, 7, 11]), (O.MODE_OR, [2, 5, 19]),
                   (O.MODE_PHRASE, [0, 1, 2]), (O.MODE_OR, [17]), (O.MODE_AND, [4, 25])]  # 25: no such ordinal
        scores, _, dd, counts = dev.search(queries, 10)
        for i, (mode, terms) in enumerate(queries):
            if 25 in terms:
                assert counts[i] == 0
                continue
            want = O.search(seg, terms, mode, 10, pruned=False)
            got = [(float(scores[i, j]), int(dd[i, j])) for j in range(int(counts[i]))]
            assert [d for _, d in got] == [d for _, d in want]
            for (gs, _), (ws, _) in zip(got, want):
                assert abs(gs - ws) <= 1e-5 * abs(ws)
    finally:
        dev.close()
