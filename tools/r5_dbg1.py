import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as O
import tantivy_amd as ta
seed=1
seg = O.synth_segment(300_000, n_terms=48, segment_ord=seed)
qs = [(O.MODE_AND, t.tolist()) for t in O.zipf_queries(3000, 2, 48, seed=100 + seed)]
def run(env=None):
    dev = ta.DeviceIndex([seg]); dev.set_option("ashare_min_batch", 32)
    try:
        dev.set_option("exhaustive", 0)
        sc, _, dc, ct = dev.search(qs, 10)
        dev.set_option("exhaustive", 1)
        se, _, de, ce = dev.search(qs, 10)
    finally:
        dev.close()
    bad = [i for i in range(len(qs)) if not (np.array_equal(dc[i], de[i]) and np.array_equal(sc[i], se[i]))]
    return bad, (sc, dc, ct, se, de, ce)
bad, r = run()
print("mode", os.environ.get("TQ_AS_BOUND"), "bad", len(bad))
sc, dc, ct, se, de, ce = r
for i in bad[:5]:
    print(qs[i], [seg.terms[t].doc_freq for t in qs[i][1]])
    print("  got ", list(zip(sc[i][:ct[i]].tolist(), dc[i][:ct[i]].tolist())))
    print("  want", list(zip(se[i][:ce[i]].tolist(), de[i][:ce[i]].tolist())))
