#!/usr/bin/env python
"""bench.py — headline benchmark: 2-term AND + BM25 top-10 QPS on a 10M-doc synthetic Zipf
segment per GPU (BASELINE.json configs[1]), with the HBM roofline of the scan
kernel and the CPU restatement of tantivy's own executor timed beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of queries (posting lists resident in HBM;
query weights prepared outside the timed region, like `query.weight()` in tantivy's benches).
The timed mode is the reference's own: block-max pruned top-k (block_wand_intersection).  The
exhaustive mode (every match scored) is run beside it: every query's top-k must be identical in
both (the run aborts otherwise) and its roofline is reported as `roofline_other_mode`.
Roles of oracle/ here: (1) workload generator — it serialises the synthetic index into tantivy's
byte format before anything is timed; (2) the cpu_baseline leg; (3) a post-hoc parity spot check.
The timed GPU leg runs only tantivy_amd (HIP kernels + C ABI + C++ host mirror).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--workload", default="and2", choices=["and2", "or5", "phrase3", "mixed", "bool"])
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-queries", type=int, default=200)
    ap.add_argument("--exhaustive", action="store_true",
                    help="time the exhaustive mode (score every match) instead of the reference's "
                         "block-max pruned execution; both are always run and compared")
    ap.add_argument("--pruned", action="store_true", help="(default; kept for old command lines)")
    return ap.parse_args()


def build_queries(O, workload, n, k):
    if workload == "and2":
        ids = O.zipf_queries(n, 2, 256, seed=20260921)
        return [(O.MODE_AND, q.tolist()) for q in ids], k or 10
    if workload == "or5":
        ids = O.zipf_queries(n, 5, 256, seed=20260922)
        return [(O.MODE_OR, q.tolist()) for q in ids], k or 100
    if workload == "phrase3":
        rng = np.random.default_rng(20260923)
        starts = rng.integers(0, 30, size=n)
        return [(O.MODE_PHRASE, [int(s), int(s) + 1, int(s) + 2]) for s in starts], k or 10
    if workload == "bool":
        # the shapes of the reference's union_intersection group (benches/and_or_queries.rs:150-153):
        # `+c +(b OR d)`, `+e +(c OR a)`, `+(c OR b) +(d OR e)`, plus `+a b -c`
        import tantivy_amd as T
        ids = O.zipf_queries(n, 4, 256, seed=20260924)
        M, S, N = T.MUST, T.SHOULD, T.MUST_NOT
        shapes = [(3, [M, M, M], [0, 1, 1]), (4, [M, M, M, M], [0, 0, 1, 1]),
                  (3, [M, S, N], None), (3, [M, M, M], [0, 0, 1])]
        qs = []
        for i, q in enumerate(ids):
            nt, occ, cof = shapes[i % len(shapes)]
            qs.append((T.MODE_BOOL, q.tolist()[:nt], occ, cof, 0))
        return qs, k or 10
    a = O.zipf_queries(n // 2, 2, 256, seed=20260921)
    o = O.zipf_queries(n - n // 2, 5, 256, seed=20260922)
    qs = []
    for i in range(n):
        qs.append((O.MODE_AND, a[i // 2].tolist()) if i % 2 == 0 else (O.MODE_OR, o[i // 2].tolist()))
    return qs, k or 10


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from tantivy_amd import build as product_build

    if not os.path.exists(product_build.LIB):  # normally prebuilt (__graft_entry__.build())
        if rank == 0:
            product_build.build()
        if dist is not None:
            dist.barrier()
    from oracle import oracle as O  # workload generator + cpu baseline + spot check only
    import tantivy_amd
    from tantivy_amd import distributed as D

    with_pos = args.workload == "phrase3"
    t0 = time.time()
    seg = O.synth_segment(args.docs, n_terms=256, segment_ord=rank, with_positions=with_pos,
                          phrase_terms=32)
    t_gen = time.time() - t0
    dev = tantivy_amd.DeviceIndex([seg], devices=[local_rank])
    dev.set_option("timing", 1)
    for name in ("dense_ratio", "dense_budget_x"):  # experiments: TQ_OPT_dense_ratio=...
        if os.environ.get("TQ_OPT_" + name):
            dev.set_option(name, int(os.environ["TQ_OPT_" + name]))
    pruned_mode = not args.exhaustive
    my_stats = (seg.max_doc, seg.total_num_tokens, [t.doc_freq for t in seg.terms])
    all_stats = [my_stats]
    if world > 1:
        all_stats = [None] * world
        dist.all_gather_object(all_stats, my_stats)
        for r, st in enumerate(all_stats):
            if r != rank:
                dev.add_remote_stats(*st)
    queries, k = build_queries(O, args.workload, args.queries, args.k)
    n_q = len(queries)
    dev.prepare(queries)  # Query::weight: global BM25 statistics, executor choice

    # a non-default stream: the library's kernels, the RCCL all-gather and the merge kernel are
    # all ordered on it (the legacy null stream would not order against the library's own stream)
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    d_scores = torch.empty((n_q, k), dtype=torch.float32, device="cuda")
    d_docs = torch.empty((n_q, k), dtype=torch.int32, device="cuda")
    d_counts = torch.empty(n_q, dtype=torch.int32, device="cuda")
    h_out = [torch.empty((n_q, k), dtype=torch.float32).pin_memory(),
             torch.empty((n_q, k), dtype=torch.int32).pin_memory(),
             torch.empty((n_q, k), dtype=torch.int32).pin_memory(),
             torch.empty(n_q, dtype=torch.int32).pin_memory()]

    def enqueue():
        """collect_segment on this rank's segment -> (all-gather) -> merge_top_k -> host copies.
        Nothing here waits for the GPU: the host plans batch i+1 while batch i runs."""
        dev.collect_segment_prepared_device(0, k, d_scores, d_docs, d_counts, stream)
        if world > 1:
            g = D.allgather_topk(d_scores, d_docs, d_counts)
        else:
            g = (d_scores.unsqueeze(0), d_docs.unsqueeze(0), d_counts.unsqueeze(0))
        m = D.merge_gathered_device(dev.ctx, local_rank, g[0], g[1], g[2], 0, k, stream)
        for h, t in zip(h_out, m):
            h.copy_(t, non_blocking=True)

    def step():
        enqueue()
        torch.cuda.synchronize()
        return dev.last_batch_stats()

    # untimed reference pass in the OTHER mode: every query's top-k must be identical with and
    # without block-max pruning (full-size parity property), and the exhaustive pass counts the
    # matches that enter the algorithmic-bytes figure
    def run_mode(exhaustive, n):
        dev.set_option("exhaustive", 1 if exhaustive else 0)
        kms, st = [], None
        for _ in range(n):
            st = step()
            kms.append(st["kernel_ms"])
        torch.cuda.synchronize()
        return [h.clone().numpy() for h in h_out], st, kms

    exh_out, exh_st, _ = run_mode(True, 1)
    _, _, exh_kms = run_mode(True, max(1, min(args.steps, 3)))
    prn_out, prn_st, _ = run_mode(False, 1)
    _, _, prn_kms = run_mode(False, max(1, min(args.steps, 3)))
    mode_parity = all(np.array_equal(a, b) for a, b in zip(exh_out, prn_out))
    if not mode_parity:
        n_diff = int(np.sum(np.any(exh_out[2] != prn_out[2], axis=1)))
        raise SystemExit("pruned and exhaustive results differ on %d queries" % n_diff)
    full_matches = exh_st["matches"]
    algo_bytes_full = exh_st["algorithmic_bytes"]  # postings + positions + matches + 8k
    dev.set_option("exhaustive", 0 if pruned_mode else 1)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    dev.last_batch_stats()  # start a fresh timing window
    trace = [] if os.environ.get("BENCH_TRACE") else None
    for _ in range(args.steps):
        enqueue()  # steps are pipelined: one synchronisation closes the timed region
        if trace is not None:
            trace.append(time.perf_counter() - t_start)
    torch.cuda.synchronize()
    if trace is not None:
        trace.append(time.perf_counter() - t_start)
        print("enqueue returns / final sync (ms):", [round(x * 1e3, 2) for x in trace], file=sys.stderr)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final = [h.clone().numpy() for h in h_out]
    st = dev.last_batch_stats()  # HIP-event kernel time, mean over the timed steps (<= last 16)
    kernel_ms = [st["kernel_ms"]]
    algo_bytes, matches = algo_bytes_full, st["matches"]

    # ---- single-query latency (p50), outside the timed region
    lat = []
    if rank == 0 and args.latency_queries > 0:
        for i in range(min(args.latency_queries, n_q)):
            dev.prepare([queries[i]])
            t1 = time.perf_counter()
            dev.search_prepared(k)
            lat.append(time.perf_counter() - t1)
        dev.prepare(queries)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- post-hoc parity spot check against the oracle (N=1 only: local stats == global)
    parity_checked = 0
    if world == 1:
        for i in list(range(0, n_q, max(1, n_q // 16)))[:16]:
            mode, terms = queries[i][0], queries[i][1]
            if args.workload == "bool":
                want = O.bool_search(seg, terms, queries[i][2], k, queries[i][3], queries[i][4])
            else:
                want = O.search(seg, terms, mode, k, pruned=False)
            got = [(float(final[0][i, j]), int(final[2][i, j])) for j in range(int(final[3][i]))]
            assert len(got) == len(want), (i, got, want)
            for (gs, gd), (ws, wd) in zip(got, want):
                assert gd == wd and abs(gs - ws) <= 1e-5 * abs(ws), (i, got, want)
            parity_checked += 1

    # ---- CPU baseline: the oracle's restatement of tantivy's block-WAND executors
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        specs, done, wall_total = [], 0, 0.0
        chunk = max(64, cores * 16)
        def spec_of(q):
            if args.workload == "bool":  # the restated generic scorer tree (no block-max executor)
                return O.bool_spec(seg, q[1], q[2], q[3], q[4], k)
            return O.QuerySpec(seg, q[1], O.default_weights(seg, q[1], q[0]), q[0], k,
                               list(range(len(q[1]))) if q[0] == O.MODE_PHRASE else None)

        while wall_total < args.cpu_seconds and done < n_q:
            part = queries[done:done + chunk]
            sp = [spec_of(q) for q in part]
            wall, _, _ = O.baseline_run(seg, sp, cores)
            wall_total += wall
            done += len(part)
        sp1 = [spec_of(q) for q in queries[:48]]
        _, lat1, _ = O.baseline_run(seg, sp1, 1)
        cpu = {"value": round(done / wall_total, 2), "unit": "queries/s", "cores": cores,
               "kind": "port",
               "sample": "first %d queries of the same stream, query-level parallelism on %d "
                         "threads, %.1f s; C restatement of tantivy's %s (oracle/), not the tantivy "
                         "binary" % (done, cores, wall_total,
                                     "generic scorer tree (Intersection / BufferedUnionScorer / "
                                     "RequiredOptionalScorer / Exclude under for_each_pruning_scorer)"
                                     if args.workload == "bool" else
                                     "block_wand_intersection/block_wand"),
               "p50_latency_ms_1core": round(float(np.median(lat1)) * 1e3, 3)}

    # HBM traffic per launch from the committed rocprofv3 PMC run of this same command
    traffic, traffic_note = None, "no PMC run recorded"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        key = "%s_%s_%d" % (args.workload, "pruned" if pruned_mode else "exhaustive", args.docs)
        if key in tj:
            traffic = tj[key]["hbm_bytes_per_launch"]
            traffic_note = tj[key]["note"]

    def roof(kms):
        k = float(np.mean(kms))
        a = algo_bytes_full / (k * 1e-3) / 1e9 if k > 0 else 0.0
        return {"mode": None, "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 4), "kernel_ms_avg": round(k, 4)}

    other = roof(exh_kms if pruned_mode else prn_kms)
    other["mode"] = "exhaustive" if pruned_mode else "pruned"
    total_units = n_q * args.steps * world  # one unit = one query evaluated on one segment
    value = total_units / elapsed
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
    achieved = (algo_bytes / (k_ms * 1e-3) / 1e9) if k_ms > 0 else 0.0
    out = {
        "metric": "queries_per_sec_2term_AND_bm25_top10" if args.workload == "and2"
        else "queries_per_sec_" + args.workload,
        "value": round(value, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32+f32",
        "data": "synthetic",
        "config": {
            "workload": "%s: %d queries/batch, k=%d, one %dM-doc Zipf segment per GPU (256 terms, "
                        "df_r=0.5N/r, WithFreqs%s); term ranks ~ Zipf(1)" %
                        (args.workload, n_q, k, args.docs // 1_000_000,
                         "AndPositions" if with_pos else ""),
            "unit_note": "one unit = one query evaluated on one segment; at N GPUs every query "
                         "runs on N segments (N x %dM docs) and the per-segment top-k are "
                         "all-gathered over RCCL and merged" % (args.docs // 1_000_000),
            "timed_region": "K x [collect_segment (plan + H2D of query descriptors + scan + merge "
                            "kernels) -> all-gather -> merge_top_k -> D2H], enqueued back to back "
                            "(host planning of step i+1 overlaps the GPU work of step i), one "
                            "synchronisation at each end",
            "mode": "block-max pruned (block_wand_intersection semantics)" if pruned_mode
                    else "exhaustive (every match scored)",
            "index_bytes": int(seg.idx_len),
            "index_build_s": round(t_gen, 2),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "kernel": "and_kernel" if args.workload == "and2" else args.workload + " scan kernels",
            "kernel_ms_avg": round(k_ms, 4),
            "algorithmic_bytes_per_launch": int(algo_bytes),
            "docs_scored_per_launch": int(matches),
            "matches_per_launch": int(full_matches),
            "traffic_note": traffic_note,
        },
        "roofline_other_mode": other,
        "pruned_equals_exhaustive": bool(mode_parity),
        "cpu_baseline": cpu,
        "p50_latency_ms": round(float(np.median(lat)) * 1e3, 4) if lat else None,
        "parity_checked_queries": parity_checked,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
