#!/usr/bin/env python
"""Soak run of the nested-query kernel (tree_kernel): SEEDS random configurations — segment size,
vocabulary, which lists get bitmaps, probe-pool budget, k — each every shape of tests/tree_shapes.py (SHAPES,
PHRASE_SHAPES where the segment has positions, DEEP_SHAPES) over random terms, pruned == exhaustive bit for bit and
every query against the numpy tree oracle (doc ids exact up to near-ties across the k-th rank, scores within 1e-5).
    SEEDS=40 FIRST=0 python tools/soak_tree.py"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd as ta  # noqa: E402
from tests.helpers import rel_close  # noqa: E402
from tests import tree_shapes as TS  # noqa: E402


def check(dev, seg, specs, k):
    queries = [TS.to_device(ta, sp, msm) for sp, msm in specs]
    out = {}
    for mode in (0, 1):
        dev.set_option("exhaustive", mode)
        out[mode] = dev.search(queries, k)
        st = dev.last_batch_stats()
        assert st["kernel_mask"] & ta.binding.KERNEL_TREE, st
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b), "pruned != exhaustive"
    sc, _, dc, ct = out[0]
    for i, (sp, msm) in enumerate(specs):
        want = O.tree_search(seg, TS.to_oracle(sp), k, msm, None)
        got = [(float(sc[i, j]), int(dc[i, j])) for j in range(int(ct[i]))]
        assert len(got) == len(want), (sp, msm, got, want)
        assert all(rel_close(a, b, 1e-5) for (a, _), (b, _) in zip(got, want)), (sp, msm, got, want)
        if got:
            kth = got[-1][0]
            for d in set(d for _, d in got) ^ set(d for _, d in want):
                s_ = [x for x, dd in got + want if dd == d][0]
                assert rel_close(s_, kth, 1e-5), (sp, msm, got, want)


def main():
    n_seeds = int(os.environ.get("SEEDS", "20"))
    first = int(os.environ.get("FIRST", "0"))
    bad = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(7000 + seed)
        n_docs = int(rng.choice([60_000, 150_000, 300_000, 700_000]))
        vocab = int(rng.choice([24, 48, 256, 1024]))
        with_pos = vocab <= 48 and n_docs <= 300_000
        k = int(rng.choice([1, 10, 10, 100]))
        opts = {"dense_ratio": int(rng.choice([8, 32, 64, 512])), "probe_budget_x": int(rng.choice([1, 2, 16]))}
        t0 = time.time()
        try:
            seg = O.synth_segment(n_docs, n_terms=vocab, with_positions=with_pos, phrase_terms=min(vocab, 24))
            shapes = list(TS.SHAPES) + list(getattr(TS, "DEEP_SHAPES", []))
            if with_pos:
                shapes += list(TS.PHRASE_SHAPES)
            specs = []
            for shape, msm in shapes:
                for _ in range(2):
                    hi = min(vocab, int(rng.choice([16, 40, 200, 1000])))
                    ids = rng.permutation(hi)[:8].tolist() if hi >= 8 else rng.permutation(vocab)[:8].tolist()
                    specs.append((shape(ids), msm))
            dev = ta.DeviceIndex([seg])
            try:
                for name, v in opts.items():
                    dev.set_option(name, v)
                check(dev, seg, specs, k)
                ev = dev.segment_stats(0).get("probe_evictions", 0)
            finally:
                dev.close()
            print("seed %d ok: %d docs, %d terms, k %d, %d nested queries, %s, probe evictions %s, %.1f s" %
                  (seed, n_docs, vocab, k, len(specs), opts, ev, time.time() - t0), flush=True)
        except Exception:
            bad += 1
            print("seed %d FAILED: %d docs, %d terms, k %d, %s" % (seed, n_docs, vocab, k, opts), flush=True)
            traceback.print_exc(limit=3)
            sys.stdout.flush()
            if os.environ.get("STOP", "1") != "0":
                break
    print("soak: %d seeds, %d failed" % (n_seeds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
