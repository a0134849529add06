#!/usr/bin/env python
"""In-kernel region timers (TQ_DEBUG bits 16..19, union and phrase kernels): wave cycles per region.
Union kernel regions: 1 whole chunk, 2 query setup, 3 flush, 4 tile bookkeeping + threshold,
5 pre-filter, 6 bitmap sweep (B / C inside included), 7 decode path (B / C included), 8 stage B,
9 stage C.
usage: python tools/probe_phases.py <or5|phrase3> ; runs once per phase (the debug word is read at
library load)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2:  # child: one phase
    from oracle import oracle as O
    import numpy as np
    import tantivy_amd

    wl = sys.argv[1]
    seg = O.synth_segment(10_000_000, n_terms=256, with_positions=wl == "phrase3", phrase_terms=32)
    dev = tantivy_amd.DeviceIndex([seg], devices=[0])
    dev.set_option("timing", 1)
    dev.set_option("exhaustive", 0)
    if wl == "or5":
        qs = [(O.MODE_OR, q.tolist()) for q in O.zipf_queries(1000, 5, 256, seed=20260922)]
        k = 100
    elif wl == "bool":
        import tantivy_amd as T
        M, S, N = T.MUST, T.SHOULD, T.MUST_NOT
        shapes = [(3, [M, M, M], [0, 1, 1]), (4, [M, M, M, M], [0, 0, 1, 1]), (3, [M, S, N], None), (3, [M, M, M], [0, 0, 1])]
        qs = []
        for i, q in enumerate(O.zipf_queries(2000, 4, 256, seed=20260924)):
            nt, occ, cof = shapes[i % 4]
            qs.append((T.MODE_BOOL, q.tolist()[:nt], occ, cof, 0))
        k = 10
    elif wl == "and2":
        qs = [(O.MODE_AND, q.tolist()) for q in O.zipf_queries(10000, 2, 256, seed=20260921)]
        k = 10
    else:
        starts = np.random.default_rng(20260923).integers(0, 30, size=1000)
        qs = [(O.MODE_PHRASE, [int(s), int(s) + 1, int(s) + 2]) for s in starts]
        k = 10
    dev.prepare(qs)
    for _ in range(2):
        dev.search_prepared(k)
        st = dev.last_batch_stats()
    print("phase %s kernel %.3f ms wave-cycles %.4g" % (sys.argv[2], st["kernel_ms"], st["matches"] * 16.0))
    dev.close()
else:
    if sys.argv[1] in ("or5", "bool") and not os.environ.get("TQ_LIB_PATH"):  # the union kernel's timers are compiled out by default
        lib = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "timers",
                              "tq_union.hip", "-DTQ_U_TIMERS=1"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
        os.environ["TQ_LIB_PATH"] = lib
    for ph in range(1, 10):
        env = dict(os.environ, TQ_DEBUG=str((ph << 16) | int(os.environ.get("TQ_DEBUG_BASE", "0"))))
        out = subprocess.run([sys.executable, __file__, sys.argv[1], str(ph)], env=env,
                             capture_output=True, text=True).stdout
        print(out.strip().splitlines()[-1] if out.strip() else "phase %d: no output" % ph)
