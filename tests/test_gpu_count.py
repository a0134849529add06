"""GPU parity of the Count collector over bitmaps (tantivy_amd/csrc/tq_count.hip): a query whose lists all
have a bitmap is a bitwise expression over bitmap words, its count the popcount — against the oracle's doc
sets and against the exhaustive scan (Count = src/collector/count_collector.rs:39-80; the reference counts a
union out of bitset words the same way, src/query/union/buffered_union.rs:331-351).  Deletes are the AliveBitSet
ANDed in (alive_bitset.rs:58-61)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_gpu_bshare import SHAPES
from tests.test_gpu_round3 import _alive_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def _want(seg, q, deleted):
    if q[0] == O.MODE_BOOL:
        d, _ = O.bool_match_all(seg, q[1], q[2], q[3], q[4])
    else:
        d, _ = O.match_all(seg, q[1], q[0])
    return sum(1 for doc in d.tolist() if doc not in deleted)


@pytest.mark.parametrize("seed", [31, 32])
def test_counts_from_bitmaps_equal_the_scan_and_the_oracle(ta, seed):
    rng = np.random.default_rng(seed)
    seg = O.synth_segment(100_000 + 999 * seed, n_terms=48, with_positions=False)
    queries = []
    for occ, cof, msm in SHAPES * 2 + [([O.MUST, O.SHOULD, O.SHOULD], None, 2), ([O.SHOULD, O.SHOULD, O.SHOULD], [0, 0, 1], 1),
                                       ([O.MUST, O.SHOULD], None, 2), ([O.MUST_NOT, O.MUST_NOT], None, 0)]:
        terms = rng.choice(48, size=len(occ), replace=False).tolist()
        queries.append((ta.MODE_BOOL, terms, list(occ), cof, msm))
    for n in (1, 2, 3, 5):
        for _ in range(6):
            terms = rng.choice(48, size=n, replace=False).tolist()
            queries.append((O.MODE_AND, terms))
            queries.append((O.MODE_OR, terms))
    deleted = set(rng.choice(seg.max_doc, size=seg.max_doc // 7, replace=False).tolist())
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 4096)    # every list gets a bitmap
        dev.set_option("dense_budget_x", 256)
        for dels in ((), deleted):
            dev.set_alive_bitset(_alive_bytes(seg.max_doc, dels) if dels else None)
            want = [_want(seg, q, dels) for q in queries]
            dev.set_option("count_bitmap_ratio", 1 << 30)  # always
            got = dev.count(queries)
            st = dev.last_batch_stats()
            assert st["kernel_mask"] & ta.binding.KERNEL_COUNT_BITMAPS, st
            assert got.tolist() == want, [(q, g, w) for q, g, w in zip(queries, got.tolist(), want) if g != w][:5]
            dev.set_option("count_bitmap_ratio", 0)  # never
            scan = dev.count(queries)
            st = dev.last_batch_stats()
            assert not (st["kernel_mask"] & ta.binding.KERNEL_COUNT_BITMAPS), st
            assert scan.tolist() == want
            dev.set_option("count_bitmap_ratio", 128)  # the default: long driving clauses from bitmaps, rare leaders scanned
            assert dev.count(queries).tolist() == want
    finally:
        dev.close()


def test_lists_without_bitmaps_get_one_for_the_batch(ta):
    """dense_ratio 8: three lists of 48 have a bitmap; every other list a query names is scattered into the
    batch's scratch as plain bit words (count_scatter_kernel) and the query counted from bitmaps all the same."""
    rng = np.random.default_rng(77)
    seg = O.synth_segment(90_000, n_terms=48, with_positions=False)
    queries = []
    for occ, cof, msm in SHAPES:
        terms = rng.choice(48, size=len(occ), replace=False).tolist()
        queries.append((ta.MODE_BOOL, terms, list(occ), cof, msm))
    for n in (1, 2, 4, 7):
        for _ in range(5):
            terms = rng.choice(48, size=n, replace=False).tolist()
            queries += [(O.MODE_AND, terms), (O.MODE_OR, terms)]
    deleted = set(rng.choice(seg.max_doc, size=seg.max_doc // 9, replace=False).tolist())
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 8)
        dev.set_option("count_bitmap_ratio", 1 << 30)
        for dels in ((), deleted):
            dev.set_alive_bitset(_alive_bytes(seg.max_doc, dels) if dels else None)
            got = dev.count(queries)
            st = dev.last_batch_stats()
            # (the "m of n" Should shapes are not bitwise expressions: they are scanned by the boolean kernel)
            assert st["kernel_mask"] & ta.binding.KERNEL_COUNT_BITMAPS, st
            assert not (st["kernel_mask"] & ~(ta.binding.KERNEL_COUNT_BITMAPS | ta.binding.KERNEL_BOOL)), st
            want = [_want(seg, q, dels) for q in queries]
            assert got.tolist() == want, [(q, g, w) for q, g, w in zip(queries, got.tolist(), want) if g != w][:5]
        assert dev.segment_stats(0)["n_dense_lists"] <= 4
    finally:
        dev.close()


def test_rare_leaders_are_scanned(ta):
    seg = O.synth_segment(120_000, n_terms=64, with_positions=False)
    queries = [(O.MODE_OR, [0, 1, 2]), (O.MODE_OR, [0, 60, 2]), (O.MODE_AND, [0, 1]), (O.MODE_AND, [63, 62]),
               (O.MODE_PHRASE, [0, 1]) if False else (O.MODE_OR, [5])]
    dev = ta.DeviceIndex([seg])
    try:
        dev.set_option("dense_ratio", 16)  # lists 0..6 get a bitmap
        got = dev.count(queries)
        st = dev.last_batch_stats()
        assert st["kernel_mask"] & ta.binding.KERNEL_COUNT_BITMAPS and st["kernel_mask"] != ta.binding.KERNEL_COUNT_BITMAPS, st
        assert got.tolist() == [_want(seg, q, ()) for q in queries]
    finally:
        dev.close()
