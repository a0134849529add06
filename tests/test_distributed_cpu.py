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


class _ThreeGatherComm:
    """Stands in for tantivy_amd.distributed.Comm on the CPU: tq_allgather_topk's layout
    (csrc/tq_comm.cpp:163-185) — THREE all-gathers, scores / docs / counts each into its own
    [world][rows][stride] slab — over gloo instead of one grouped RCCL launch."""

    def allgather_topk(self, scores, docs, counts, out, stream):
        dist.all_gather_into_tensor(out[0].view(-1), scores.contiguous().view(-1))
        dist.all_gather_into_tensor(out[1].view(-1), docs.contiguous().view(-1))
        dist.all_gather_into_tensor(out[2].view(-1), counts.contiguous().view(-1))


def _worker_three_gather(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from tantivy_amd import distributed as D

        S, K = 2, 6  # two local segments per rank: the [world][S * n][k] -> [world * S][n][k] view matters
        segs = [O.synth_segment(20_000, n_terms=16, segment_ord=o) for o in range(world * S)]
        nd = sum(s.max_doc for s in segs)
        nt = sum(s.total_num_tokens for s in segs)
        queries = [(O.MODE_AND, [0, 1]), (O.MODE_OR, [2, 5, 9]), (O.MODE_AND, [3, 15]), (O.MODE_OR, [15]),
                   (O.MODE_AND, [1, 2, 4])]
        n = len(queries)

        def weights(seg, mode, terms):
            dfs = [sum(s.terms[t].doc_freq for s in segs) for t in terms]
            return O.default_weights(seg, terms, mode, total_num_docs=nd, total_num_tokens=nt, dfs=dfs)

        # what ShardRunner.enqueue's collect_segment calls leave in `local`: [S * n, k] slabs, local
        # segment j's rows at j * n (here from the oracle: no GPU)
        scores = np.zeros((S * n, K), np.float32)
        docs = np.full((S * n, K), 0x7FFFFFFF, np.uint32)
        counts = np.zeros(S * n, np.uint32)
        for j in range(S):
            seg = segs[rank * S + j]
            for qi, (mode, terms) in enumerate(queries):
                hits = O.search(seg, terms, mode, K, weights=weights(seg, mode, terms), pruned=True)
                counts[j * n + qi] = len(hits)
                for x, (s, d) in enumerate(hits):
                    scores[j * n + qi, x], docs[j * n + qi, x] = s, d
        local = (torch.from_numpy(scores), torch.from_numpy(docs.view(np.int32)), torch.from_numpy(counts.view(np.int32)))
        gathered = (torch.empty((world, S * n, K), dtype=torch.float32), torch.empty((world, S * n, K), dtype=torch.int32),
                    torch.empty((world, S * n), dtype=torch.int32))
        g = D.exchange_topk(local, n, K, S, world, comm=_ThreeGatherComm(), gathered=gathered)
        assert tuple(g[0].shape) == (world * S, n, K) and tuple(g[2].shape) == (world * S, n)
        out_s, out_o, out_d, out_c = D.merge_gathered_host(g[0], g[1], g[2], 0, K)
        for qi, (mode, terms) in enumerate(queries):
            allhits = []
            for o, s in enumerate(segs):
                allhits += [(sc, o, d) for sc, d in O.search(s, terms, mode, K, weights=weights(s, mode, terms))]
            want = O.merge_top_k(allhits, 0, K)
            got = [(float(out_s[qi, i]), int(out_o[qi, i]), int(out_d[qi, i])) for i in range(int(out_c[qi]))]
            assert got == want, (rank, qi, got, want)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_three_gather_exchange_world2_two_segments_per_rank():
    """ShardRunner.enqueue's exchange step (distributed.exchange_topk) with world = 2 and S = 2 local
    segments per rank, through a communicator with tq_allgather_topk's three-gather layout: the
    merged result of the 4-segment index must be the oracle's on every rank."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker_three_gather, args=(r, 2, port, ret)) for r in range(2)]
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
