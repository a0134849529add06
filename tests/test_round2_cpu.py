"""CPU-side checks added in round 2: the hand-written BitPacker4x packer of tests/helpers.py against
the oracle's, legacy (non-strict) posting lists through the oracle's reader, and the NaN-safe
ordering of tq_merge_topk."""
import numpy as np

from oracle import oracle as O
from tests.helpers import legacy_posting_list, pack4x, random_postings


def test_pack4x_helper_matches_oracle_packer():
    rng = np.random.default_rng(5)
    for b in (0, 1, 5, 13, 31, 32):
        vals = (rng.integers(0, 1 << b, size=128, dtype=np.uint64) if b else np.zeros(128, np.uint64))
        vals[-1] = (1 << b) - 1 if b else 0
        nb, data = O.compress_block_unsorted(vals.astype(np.uint32), False)
        assert nb == b and bytes(data) == pack4x(vals.tolist(), b)


def test_oracle_reads_legacy_non_strict_lists():
    """compression/mod.rs:105-125: readers accept blocks without the strict-delta flag."""
    rng = np.random.default_rng(31)
    md = 100_000
    for df in (128, 129, 128 * 5 + 77):
        old = random_postings(rng, md, df, max_tf=9)
        old[0] = (0, 3)
        old = sorted(dict(old).items())
        legacy = legacy_posting_list(old)
        body = (0).to_bytes(8, "little") + legacy
        seg = O.Segment(md, O.WITH_FREQS, np.frombuffer(body, np.uint8), np.zeros(0, np.uint8), None,
                        [O.TermInfo(len(old), 0, len(legacy), 0, 0)], 0)
        docs, tfs = O.decode_postings(seg, 0)
        assert docs.tolist() == [d for d, _ in old] and tfs.tolist() == [t for _, t in old]


def test_merge_topk_is_a_total_order_even_with_nan():
    """the comparator must be a strict weak order whatever the floats are."""
    from tantivy_amd import binding as B

    rng = np.random.default_rng(3)
    S, n, k = 3, 50, 16
    sc = rng.random((S, n, k)).astype(np.float32)
    sc[0, :, 3] = np.nan
    sc[1, :, 5] = -0.0
    sc[2, :, 5] = 0.0
    dc = rng.integers(0, 1000, size=(S, n, k)).astype(np.uint32)
    ct = np.full((S, n), k, np.uint32)
    out_s = np.zeros((n, 10), np.float32)
    out_o = np.zeros((n, 10), np.uint32)
    out_d = np.zeros((n, 10), np.uint32)
    out_c = np.zeros(n, np.uint32)
    B._check(B.lib().tq_merge_topk(B._f32(sc), B._u32(dc), B._u32(ct), S, n, k, 0, 10, B._f32(out_s),
                                   B._u32(out_o), B._u32(out_d), B._u32(out_c)))
    assert np.all(out_c == 10)
    fin = out_s[:, 1:]  # NaN (positive payload) sorts above every number: at most rank 0
    assert not np.isnan(fin).any()
    assert np.all(fin[:, :-1] >= fin[:, 1:])
