"""world_size-2 gloo test of the multi-GPU path's only exchange step: per-segment top-k
all-gather + merge_top_k.  Per-rank segment results come from the oracle here
(no GPU); the gather/merge code is the product's (tantivy_amd/distributed.py + tq_merge_topk)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from tantivy_amd import distributed as D

        K, OFFSET, LIMIT = 7, 2, 5
        segs = [O.synth_segment(30_000, n_terms=16, segment_ord=o) for o in range(world)]
        nd = sum(s.max_doc for s in segs)
        nt = sum(s.total_num_tokens for s in segs)
        queries = [(O.MODE_AND, [0, 1]), (O.MODE_OR, [2, 5, 9]), (O.MODE_AND, [3, 15]),
                   (O.MODE_OR, [15])]
        seg = segs[rank]  # this rank's segment
        scores = np.zeros((len(queries), K), np.float32)
        docs = np.full((len(queries), K), 0x7FFFFFFF, np.uint32)
        counts = np.zeros(len(queries), np.uint32)
        for qi, (mode, terms) in enumerate(queries):
            dfs = [sum(s.terms[t].doc_freq for s in segs) for t in terms]
            w = O.default_weights(seg, terms, mode, total_num_docs=nd, total_num_tokens=nt, dfs=dfs)
            hits = O.search(seg, terms, mode, K, weights=w, pruned=True)
            counts[qi] = len(hits)
            for j, (s, d) in enumerate(hits):
                scores[qi, j], docs[qi, j] = s, d
        g = D.allgather_topk(torch.from_numpy(scores), torch.from_numpy(docs.view(np.int32)),
                             torch.from_numpy(counts.view(np.int32)))
        out_s, out_o, out_d, out_c = D.merge_gathered_host(*g, OFFSET, LIMIT)
        # every rank must hold the same, oracle-identical answer
        for qi, (mode, terms) in enumerate(queries):
            dfs = [sum(s.terms[t].doc_freq for s in segs) for t in terms]
            allhits = []
            for o, s in enumerate(segs):
                w = O.default_weights(s, terms, mode, total_num_docs=nd, total_num_tokens=nt, dfs=dfs)
                allhits += [(sc, o, d) for sc, d in O.search(s, terms, mode, K, weights=w)]
            want = O.merge_top_k(allhits, OFFSET, LIMIT)
            got = [(float(out_s[qi, i]), int(out_o[qi, i]), int(out_d[qi, i]))
                   for i in range(int(out_c[qi]))]
            assert got == want, (rank, qi, got, want)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_allgather_merge_world2():
    port = _free_port()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(0) and ret.get(1)


def test_bench_launcher_spawns_ranks():
    """`python bench.py --gpus 2` without a launcher must spawn its two ranks itself and print ONE
    JSON line with n_gpus == 2 (it used to run one rank).  --selftest-launcher runs
    the exchange plumbing over gloo with synthetic per-rank results instead of GPU work."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--selftest-launcher"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j == {"selftest": "launcher", "n_gpus": 2, "ok": True}
