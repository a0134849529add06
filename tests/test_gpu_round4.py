"""GPU tests added in round 4: the per-device scratch (one set of partial / result lists and staging
lists for all segments of a device) under concurrent use, and its accounting."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_gpu_round3 import _check, _mixed_stream, _oracle_merged

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ta():
    import tantivy_amd

    return tantivy_amd


def test_segments_of_a_device_share_one_scratch(ta):
    """4 x 150k-doc segments on one GPU: the big batch buffers exist once (device_scratch_bytes is the
    same figure for every segment, the per-segment scratch stays small), results == the oracle's merge."""
    segs = [O.synth_segment(150_000, n_terms=64, segment_ord=o) for o in range(4)]
    queries = _mixed_stream(400, 64, 31)
    want = _oracle_merged(segs, {}, queries, 10)
    dev = ta.DeviceIndex(segs, devices=[0])
    try:
        for mode in (0, 1):
            dev.set_option("exhaustive", mode)
            _check(dev.search(queries, 10), want, queries)
        st = [dev.segment_stats(o) for o in range(4)]
        assert len({x["device_scratch_bytes"] for x in st}) == 1 and st[0]["device_scratch_bytes"] > 0, st
        for x in st:  # what a segment keeps for itself: staging blobs, slots, result slabs, decode scratch
            assert x["scratch_bytes"] < 64 << 20, x
    finally:
        dev.close()


def test_two_threads_two_segments_one_scratch(ta):
    """Two host threads, each searching its own segment of ONE context in a loop (raw C ABI, the
    segments' own streams): the batches take turns on the device's shared scratch; every result must
    equal the single-threaded one."""
    segs = [O.synth_segment(200_000, n_terms=64, segment_ord=o) for o in range(2)]
    qsets = [_mixed_stream(300, 64, 5), _mixed_stream(300, 64, 6)]
    cache = np.ascontiguousarray(ta.bm25_for_terms([1000], 400_000, 400_000 * 20)[1], np.float32)
    weights = [[[3.0 - 0.4 * j for j in range(len(q[1]))] for q in qs] for qs in qsets]
    dev = ta.DeviceIndex(segs, devices=[0])
    try:
        base = [dev.raw_search(qsets[o], weights[o], cache, 10, segment_ord=o) for o in range(2)]
        errors = []

        def worker(o):
            try:
                for _ in range(12):
                    got = dev.raw_search(qsets[o], weights[o], cache, 10, segment_ord=o)
                    for a, b in zip(got, base[o]):
                        if not np.array_equal(a, b):
                            errors.append((o, "mismatch"))
                            return
            except Exception as e:  # noqa: BLE001
                errors.append((o, repr(e)))

        th = [threading.Thread(target=worker, args=(o,)) for o in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        st = [dev.segment_stats(o) for o in range(2)]
        assert st[0]["device_scratch_bytes"] == st[1]["device_scratch_bytes"] > 0, st
    finally:
        dev.close()
