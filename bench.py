#!/usr/bin/env python
"""bench.py — headline benchmark: 2-term AND + BM25 top-10 QPS on a 10M-doc synthetic Zipf
segment per GPU (BASELINE.json configs[1]), with the HBM roofline of the scan
kernel and the CPU restatement of tantivy's own executor timed beside it; in the same run the
other BASELINE configs (5-term OR top-100, 3-word phrase, mixed AND/OR stream) and the
8-segment index of config 5 sharded over the GPUs of the run (strong scaling).

    python bench.py --gpus N --steps K --warmup W        # N > 1: spawns the N ranks itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # also fine

One step = one pass of the hot path over one batch of queries (posting lists resident in HBM;
query weights prepared outside the timed region, like `query.weight()` in tantivy's benches).
The timed mode is the reference's own: block-max pruned top-k (block_wand_intersection /
block_wand).  The exhaustive mode (every match scored) is run beside it: every query's top-k must
be identical in both (the run aborts otherwise).
Roles of oracle/ here: (1) workload generator — it serialises the synthetic index into tantivy's
byte format before anything is timed; (2) the cpu_baseline legs; (3) a post-hoc parity spot check.
The timed GPU legs run only tantivy_amd (HIP kernels + C ABI + C++ host mirror).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

# The GPU box shows 256 CPUs and grants 16 (cgroup cpu.max): OpenMP / BLAS pools sized by the former
# (torch: 128 threads) spin between parallel regions, burn the quota and get the WHOLE process throttled
# for the rest of a 100 ms period — seen as 10-70 ms stalls of single steps.  Nothing here needs them.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SEGMENTS_STRONG = 8  # BASELINE.json configs[4]: 80M docs in 8 segments


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # (100 x ~1 ms: the timed region of the headline is a tenth of a second)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--terms", type=int, default=256,
                    help="vocabulary of the synthetic segment and of the query stream (df_r = 0.5 N / r, "
                         "ranks ~ Zipf(1)).  256 = SURVEY.md section 8d; larger vocabularies (4096, 65536) give the "
                         "HBM-resident data point: few repeated queries, most lists without a bitmap "
                         "(implies --no-side)")
    ap.add_argument("--segments", type=int, default=1,
                    help="segments per GPU of the MAIN workload (global BM25 statistics, merge_top_k over "
                         "them): 8 puts ~1.1 GB of index + side tables behind the scan kernels, well "
                         "beyond the 256 MB Infinity Cache (implies --no-side, --no-cpu-baseline)")
    ap.add_argument("--queries", type=int, default=None,
                    help="queries per batch (default: 10000 for and2 / mixed, 1000 for or5 / phrase3 / bool)")
    ap.add_argument("--workload", default="and2", choices=["and2", "and2_distinct", "or5", "phrase3", "phrase3_adj", "mixed", "bool"])
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-queries", type=int, default=200)
    ap.add_argument("--exhaustive", action="store_true",
                    help="time the exhaustive mode (score every match) instead of the reference's "
                         "block-max pruned execution; both are always run and compared")
    ap.add_argument("--pruned", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-side", action="store_true",
                    help="only the main workload (profiling runs): skip other_workloads and strong_scaling")
    ap.add_argument("--side-steps", type=int, default=40,
                    help="timed steps of every side workload (pipelined like the main loop: with 10 steps the fill and the drain of the "
                         "pipeline were a fifth of the time)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU plumbing check of the rank launcher (gloo, no GPU work, no timing)")
    ap.add_argument("--no-pmc-inline", action="store_true",
                    help="skip the rocprofv3 passes that measure the fabric traffic of THIS run's kernels (outside the "
                         "timed region; roofline.traffic / frac then come from profiles/traffic.json if its kernel hash matches)")
    ap.add_argument("--pmc-child", default=None, help="(internal) the workload a rocprofv3 pass of --pmc-inline runs")
    ap.add_argument("--no-stream", action="store_true", help="skip the stream block (different batches, Query::weight timed)")
    ap.add_argument("--stream-vocabs", default="256,4096,65536,1048576",
                    help="vocabularies of the stream block (comma separated; each its own 10M-doc segment)")
    ap.add_argument("--stream-batches", type=int, default=24)
    ap.add_argument("--stream-serial", action="store_true",
                    help="stream block: Query::weight on the enqueueing thread (default: a second host thread prepares the next "
                         "batch while this one is planned and enqueued; profiles/r05_stream_prepare_thread.txt has both)")
    ap.add_argument("--check-queries", type=int, default=512,
                    help="queries of every workload checked against the oracle, spread over the kernel families that ran")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one
    rank per GPU (Executor::MultiThread's role, executor.rs:61-104, with processes for threads)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def selftest_launcher(rank, world):
    """World-size-N gloo run of the exchange plumbing with synthetic per-rank results (no GPU)."""
    import torch
    import torch.distributed as dist

    from tantivy_amd import distributed as D

    dist.init_process_group("gloo")
    n, k = 5, 4
    scores = torch.arange(n * k, dtype=torch.float32).reshape(n, k).flip(1) + 100.0 * rank
    docs = (torch.arange(n * k, dtype=torch.int32).reshape(n, k) + 7 * rank)
    counts = torch.full((n,), k, dtype=torch.int32)
    g = D.allgather_topk(scores, docs, counts)
    out_s, out_o, out_d, out_c = D.merge_gathered_host(*g, 0, k)
    ok = bool(np.all(out_o == world - 1) and np.all(out_c == k) and
              np.array_equal(out_s, scores.numpy() - 100.0 * rank + 100.0 * (world - 1)))
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ok": all(flags)}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


# ----------------------------------------------------------------------------- workloads
DEFAULT_QUERIES = {"and2": 10_000, "and2_distinct": 10_000, "mixed": 10_000, "or5": 1_000, "phrase3": 1_000,
                   "phrase3_adj": 1_000, "bool": 2_000}
PHRASE_TERMS = 64  # ranks with planted phrase positions in a segment built with positions
PHRASE_WORKLOADS = ("phrase3", "phrase3_adj")


def build_queries(O, workload, n, k, terms=256):
    if workload == "and2":
        ids = O.zipf_queries(n, 2, terms, seed=20260921)
        return [(O.MODE_AND, q.tolist()) for q in ids], k or 10
    if workload == "and2_distinct":
        # the headline stream with every repeated query dropped (the 10 000 Zipf draws of and2 hold 3 877
        # distinct pairs): what the shared-intersection launch does when no two queries are the same
        seen, qs, seed = set(), [], 20260921
        while len(qs) < n:
            for q in O.zipf_queries(4 * n, 2, terms, seed=seed):
                key = tuple(sorted(q.tolist()))
                if key not in seen:
                    seen.add(key)
                    qs.append((O.MODE_AND, q.tolist()))
                    if len(qs) == n:
                        break
            seed += 1
        return qs, k or 10
    if workload == "or5":
        ids = O.zipf_queries(n, 5, terms, seed=20260922)
        return [(O.MODE_OR, q.tolist()) for q in ids], k or 100
    if workload == "phrase3":
        # Three DISTINCT ranks ~ Zipf(1) over the 64 most frequent terms, in rank order, as one PhraseQuery::new_with_offset
        # (phrase_query.rs:47-70) whose term offsets are the rank differences: the generator plants rank r's first
        # position at base(doc) + r - 1 in 1/20 of the docs, so every such triple has real matches.  ~950 distinct
        # phrases per 1000 (rounds 2-5 drew 30 adjacent triples of ranks 1..32 — phrase3_adj keeps that stream: it
        # re-read 43 MB of lists out of the L2 a thousand times).
        ids = O.zipf_queries(n, 3, PHRASE_TERMS, seed=20260923)
        qs = []
        for q in ids:
            r = sorted(int(x) for x in q)
            qs.append((O.MODE_PHRASE, r, [0, r[1] - r[0], r[2] - r[0]]))
        return qs, k or 10
    if workload == "phrase3_adj":
        rng = np.random.default_rng(20260923)
        starts = rng.integers(0, 30, size=n)
        return [(O.MODE_PHRASE, [int(s), int(s) + 1, int(s) + 2]) for s in starts], k or 10
    if workload == "bool":
        # the shapes of the reference's union_intersection group (benches/and_or_queries.rs:150-153):
        # `+c +(b OR d)`, `+e +(c OR a)`, `+(c OR b) +(d OR e)`, plus `+a b -c`
        import tantivy_amd as T
        ids = O.zipf_queries(n, 4, terms, seed=20260924)
        M, S, N = T.MUST, T.SHOULD, T.MUST_NOT
        shapes = [(3, [M, M, M], [0, 1, 1]), (4, [M, M, M, M], [0, 0, 1, 1]),
                  (3, [M, S, N], None), (3, [M, M, M], [0, 0, 1])]
        qs = []
        for i, q in enumerate(ids):
            nt, occ, cof = shapes[i % len(shapes)]
            qs.append((T.MODE_BOOL, q.tolist()[:nt], occ, cof, 0))
        return qs, k or 10
    a = O.zipf_queries(n // 2, 2, terms, seed=20260921)
    o = O.zipf_queries(n - n // 2, 5, terms, seed=20260922)
    qs = []
    for i in range(n):
        qs.append((O.MODE_AND, a[i // 2].tolist()) if i % 2 == 0 else (O.MODE_OR, o[i // 2].tolist()))
    return qs, k or 10


def tantivy_amd_kernel_name(bit):
    from tantivy_amd import binding as TB

    return TB.KERNEL_NAMES.get(int(bit), hex(int(bit)))


def usable_cpus():
    """Threads this process may really use: affinity mask bounded by the cgroup CPU quota
    (os.cpu_count() ignores both: round 1 reported 256 'cores' on a 16-CPU lease)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, math.ceil(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, math.ceil(quota / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def global_stats(all_stats):
    """(total docs, total tokens, doc freq per term id) over every segment of every rank
    (Bm25Weight::for_terms statistics, bm25.rs:27-50,95-129)."""
    flat = [st for lst in all_stats for st in lst]
    nd = sum(st[0] for st in flat)
    nt = sum(st[1] for st in flat)
    dfs = np.sum(np.array([st[2] for st in flat], dtype=np.int64), axis=0)
    return nd, nt, dfs


def oracle_spec(O, seg, workload, q, k, gstats):
    """QuerySpec of one query on one segment with the index-wide Bm25Weights."""
    nd, nt, dfs = gstats if gstats is not None else (None, None, None)
    if q[0] not in (O.MODE_AND, O.MODE_OR, O.MODE_PHRASE):  # boolean shapes: the generic scorer tree
        return O.bool_spec(seg, q[1], q[2], q[3], q[4], k, None, nd, nt,
                           None if dfs is None else [int(dfs[t]) for t in q[1]])
    w = O.default_weights(seg, q[1], q[0], nd, nt, None if dfs is None else [int(dfs[t]) for t in q[1]])
    offs = None
    if q[0] == O.MODE_PHRASE:
        offs = list(q[2]) if len(q) > 2 and q[2] is not None else list(range(len(q[1])))
    return O.QuerySpec(seg, q[1], w, q[0], k, offs)


def cpu_baseline(O, segs, workload, queries, k, seconds, sweep, gstats=None):
    """The oracle's C restatement of tantivy's executors on the host cores, on a bounded sample:
    every query runs on every segment of `segs` (one task per query and segment, all granted
    threads busy — Executor::MultiThread's map over segment readers, executor.rs:61-104), with the
    index-wide Bm25Weights when there are several.  Timed twice: with the SSE2 BitPacker4x decode
    (oracle/to_simd.c — the reference's decoder is SIMD code; this is `value`) and with the
    scalar decode (`qps_scalar`)."""
    cores = usable_cpus()
    many = len(segs) > 1

    def run(threads, budget, simd):
        prev = O.set_simd(simd)
        try:
            done, wall_total = 0, 0.0
            chunk = max(32, threads * 8)
            while wall_total < budget and done < len(queries):
                part = queries[done:done + chunk]
                for seg in segs:
                    specs = [oracle_spec(O, seg, workload, q, k, gstats if many else None) for q in part]
                    wall, _, _ = O.baseline_run(seg, specs, threads)
                    wall_total += wall
                done += len(part)
            return done / wall_total, done, wall_total
        finally:
            O.set_simd(prev)

    qps, done, wall = run(cores, seconds, True)
    qps_scalar = run(cores, max(1.0, seconds / 3.0), False)[0]
    what = ("generic scorer tree (Intersection / BufferedUnionScorer / RequiredOptionalScorer / "
            "Exclude under for_each_pruning_scorer)" if workload == "bool" else
            "block_wand_intersection / block_wand / PhraseScorer")
    out = {"value": round(qps, 2), "unit": "queries/s", "cores": cores, "kind": "port",
           "qps_simd": round(qps, 2), "qps_scalar": round(qps_scalar, 2),
           "sample": "first %d queries of the same stream%s, query-level parallelism on %d threads "
                     "(sched_getaffinity bounded by the cgroup CPU quota; os.cpu_count() = %d), "
                     "%.1f s; C restatement of tantivy's %s (oracle/) with an SSE2 BitPacker4x decode "
                     "(qps_scalar: the scalar decode), -O3 -march=x86-64-v3; not the tantivy binary" %
                     (done, " on each of the %d segments (global statistics)" % len(segs) if many else "",
                      cores, os.cpu_count() or 0, wall, what)}
    if sweep:
        by_threads = {}
        for t in sorted({1, 8, 32, cores}):
            if t > cores:
                continue
            by_threads[str(t)] = round(run(t, seconds / 5.0, True)[0], 1) if t != cores else out["value"]
        out["qps_by_threads"] = by_threads
        prev = O.set_simd(True)
        try:
            _, lat1, _ = O.baseline_run(segs[0], [oracle_spec(O, segs[0], workload, q, k, None)
                                                  for q in queries[:48]], 1)
        finally:
            O.set_simd(prev)
        out["p50_latency_ms_1core"] = round(float(np.median(lat1)) * 1e3, 3)
    return out


# ----------------------------------------------------------------------------- measurement
class Cluster:
    """Rank plumbing of one bench process: control plane over gloo (statistics, the RCCL id,
    timing), data plane over the C ABI's RCCL communicator."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.comm = None
        self.torch_group = None
        self.exchange_note = "single rank: no exchange"

    def init(self, torch):
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            self.dist = dist

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def make_comm(self, ctx):
        """tq_comm_init over all ranks; falls back to torch.distributed's RCCL group when the
        C-ABI communicator cannot be built (the fallback is still RCCL over xGMI)."""
        if self.world == 1:
            return
        from tantivy_amd import distributed as D

        def exchange(raw):
            box = [raw]
            self.dist.broadcast_object_list(box, src=0)
            return box[0]

        err = None
        try:
            self.comm = D.Comm(ctx, self.local_rank, self.rank, self.world, exchange)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        errs = self.all_gather_object(err)
        if any(errs):
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            if self.rank == 0:
                print("tq_comm_init failed (%s): falling back to torch.distributed RCCL" %
                      [e for e in errs if e][0], file=sys.stderr)
            self.torch_group = self.dist.new_group(backend="nccl")
            self.exchange_note = "torch.distributed all_gather_into_tensor (RCCL), tq_comm_init failed"
        else:
            self.exchange_note = ("tq_allgather_topk: one grouped ncclAllGather per step through the "
                                  "C ABI (%s)" % self.comm.library)


def measure(cl, runner, torch, queries, k, steps, warmup, time_exhaustive=False):
    """Both modes once (parity + algorithmic bytes), then W warm-up and K timed steps of the
    chosen mode, pipelined, bracketed by barrier + synchronize; max over ranks."""
    runner.prepare(queries, k)

    def one(exhaustive, n):
        runner.set_option("exhaustive", 1 if exhaustive else 0)
        st = None
        for _ in range(n):
            runner.enqueue()
            runner.synchronize()
            st = runner.batch_stats()
        return runner.results(), st

    exh_out, exh_st = one(True, 1)
    _, exh_st2 = one(True, 2)
    prn_out, prn_st = one(False, 1)
    _, prn_st2 = one(False, 2)
    same = all(np.array_equal(a, b) for a, b in zip(exh_out, prn_out))
    runner.set_option("exhaustive", 1 if time_exhaustive else 0)
    import gc
    gc.collect()  # (before the warm-up, not between it and the timed steps: a collection evicts the planner's tables
    gc_was = gc.isenabled()  # from the caches — the first timed step then took 1.7 instead of 1.0 ms — and one inside a
    gc.disable()             # 20 ms timed region is a tenth of it)
    for _ in range(warmup):  # (pipelined like the timed steps: the first time two steps are in flight
        runner.enqueue()     # at once the runtime grows its pools — a 7-10 ms hiccup that belongs here)
    runner.synchronize()
    runner.batch_stats()  # start a fresh timing window
    cl.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(steps):
        runner.enqueue()  # steps are pipelined: one synchronisation closes the timed region
        if os.environ.get("BENCH_TRACE"):
            marks.append(time.perf_counter())
    runner.synchronize()
    if marks:  # (diagnosis: host time of every enqueue call, then the final wait)
        print("[bench] enqueue ms: %s | final wait %.3f" % (
            " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip([t0] + marks[:-1], marks)),
            (time.perf_counter() - marks[-1]) * 1e3), file=sys.stderr)
    torch.cuda.synchronize()
    cl.barrier()
    elapsed = cl.max_over_ranks(time.perf_counter() - t0)
    if gc_was:
        gc.enable()
    st = runner.batch_stats()  # HIP-event kernel time, mean over the timed steps (<= last 16)
    return {"elapsed": elapsed, "stats": st, "final": runner.results(), "mode_parity": same,
            "n_diff": int(np.sum(np.any(exh_out[2] != prn_out[2], axis=1))),
            "exh_stats": exh_st2, "prn_stats": prn_st2,
            "algo_bytes_full": exh_st["algorithmic_bytes"], "full_matches": exh_st["matches"]}


def latency_curve(dev, queries, k):
    """Latency against throughput, outside the timed region (BASELINE.json's metric is queries/sec AND
    p50 latency).  (a) One caller handing over batches of b queries, synchronously (Query::weight done
    before, like the timed steps): per-batch wall time -> QPS, p50, p99.  (b) tantivy's own call
    pattern: T host threads each calling Searcher::search with ONE query at a time
    (searcher.rs:180-238); the library coalesces the per-segment calls of concurrent threads into
    batched launches (tq_search_one) — per-query wall time, QPS, mean queries per launch."""
    def pct(v, p):
        v = sorted(v)
        return round(float(v[min(len(v) - 1, int(len(v) * p))]) * 1e3, 4)

    out = {"note": "and2 stream, k=%d, one 10M-doc segment; batch: one caller, synchronous batches (plan + H2D + "
                   "kernels + D2H per call); threads: T host threads x single-query Searcher::search calls, "
                   "coalesced by tq_search_one (submit_window_us = 100)" % k, "batch": {}, "threads": {}}
    for b in (1, 4, 16, 256, 4096, 10000):
        b = min(b, len(queries))
        dev.prepare(queries[:b])
        reps = 200 if b <= 16 else (60 if b <= 256 else 25)
        t = []
        for _ in range(reps + 3):
            t1 = time.perf_counter()
            dev.search_prepared(k)
            t.append(time.perf_counter() - t1)
        t = t[3:]
        out["batch"][str(b)] = {"qps": round(b * len(t) / sum(t), 1), "p50_ms": pct(t, 0.5), "p99_ms": pct(t, 0.99)}
    for nthreads in (1, 16, 64, 256, 1024):
        # (every thread gets 150 / 60 queries: the stream repeated where it is shorter; the clock starts when all
        # threads stand at the line — a server's request threads exist before the requests do)
        n = 400 if nthreads == 1 else nthreads * (150 if nthreads <= 64 else 60)
        run = (queries * (n // len(queries) + 1))[:n]
        dev.search_concurrent(run[:4 * nthreads], k, nthreads)  # (threads + queue warm)
        dev.submit_stats(reset=True)
        _, _, _, _, lat_ms, wall_ms = dev.search_concurrent(run, k, nthreads)
        st = dev.submit_stats()
        v = sorted(float(x) for x in lat_ms)
        out["threads"][str(nthreads)] = {
            "qps": round(n / (wall_ms * 1e-3), 1), "p50_ms": round(v[len(v) // 2], 4),
            "p99_ms": round(v[min(len(v) - 1, int(len(v) * 0.99))], 4), "queries": n,
            "launches": st["batches"], "queries_per_launch": round(st["queries"] / max(1, st["batches"]), 2),
            "max_queries_per_launch": st["max_batch"]}
    return out


def frac_of(algo_bytes, kernel_ms):
    a = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    return round(a, 1), round(a / HBM_PEAK_GBS, 4)


def load_traffic(key):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        return tj.get(key)
    return None


def traffic_fields(tj, launches, kernel_ms):
    """roofline.traffic & co. from a committed PMC run (profiles/traffic.json), only while that
    run measured THIS tree's kernels: entries carry the hash of the sources of the kernels they
    measured (tools/summarize_profile.py, tantivy_amd/build.py::kernel_hash); with another hash the
    figures are withheld, not re-printed."""
    from tantivy_amd import build as product_build

    out = {"traffic": None, "physical_frac": None, "l2_hit_rate": None, "traffic_from_commit": None,
           "traffic_matches_this_build": None, "traffic_note": "no PMC run recorded"}
    if not tj:
        return out
    out["traffic_from_commit"] = tj.get("measured_on_commit")
    # the entry names the kernels it measured: valid while THEIR sources (and the device headers) are
    # unchanged — the host planner or another kernel family moving on does not touch these bytes
    if tj.get("kernel_hash"):
        now = product_build.kernel_hash(tj.get("kernels", []))
        same = tj["kernel_hash"] == now
        what = "kernel hash %s, this tree %s" % (tj["kernel_hash"], now)
    else:
        same = tj.get("csrc_hash") == product_build.csrc_hash()
        what = "csrc hash %s, this tree %s" % (tj.get("csrc_hash"), product_build.csrc_hash())
    out["traffic_matches_this_build"] = same
    if not same:
        out["traffic_note"] = ("the committed PMC run (%s) measured other kernel sources (%s): traffic / "
                               "physical_frac withheld" % (tj.get("profile"), what))
        return out
    out["traffic"] = tj["hbm_bytes_per_launch"] * launches
    out["traffic_note"] = tj["note"]
    out["l2_hit_rate"] = tj.get("l2_hit_rate")
    if kernel_ms > 0:
        out["physical_frac"] = round(out["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


SCAN_KERNEL_RE = r"(and_kernel|union_kernel_small|union_kernel|ushare_kernel|ashare_kernel|xunion_kernel|or_kernel|phrase_sweep_kernel|phrase_kernel|tree_kernel)<"


def pmc_child(args):
    """One workload, pruned mode, S synchronous steps on one segment and nothing else: what a rocprofv3 pass of
    pmc_inline() profiles (every scan-kernel dispatch of the process belongs to one of the S steps)."""
    import torch

    torch.set_num_threads(4)
    torch.cuda.set_device(0)
    from oracle import oracle as O
    from tantivy_amd import distributed as D

    wl = args.pmc_child
    seg = O.synth_segment(args.docs, n_terms=args.terms, segment_ord=0, with_positions=wl in PHRASE_WORKLOADS, phrase_terms=PHRASE_TERMS)
    runner = D.ShardRunner([seg], 0)
    for name in ("dense_ratio", "dense_budget_x", "probe_budget_x", "docmat", "docsig", "device_prepare", "or_windows"):
        if os.environ.get("TQ_OPT_" + name):
            runner.set_option(name, int(os.environ["TQ_OPT_" + name]))
    queries, k = build_queries(O, wl, args.queries or DEFAULT_QUERIES[wl], args.k, args.terms)
    runner.prepare(queries, k)
    runner.set_option("exhaustive", 0)
    for _ in range(args.steps):
        runner.enqueue()
        runner.synchronize()
    print(json.dumps({"pmc_child": wl, "steps": args.steps, "kernels": runner.batch_stats().get("kernels")}))
    runner.close()
    return 0


def pmc_inline(args, workload, n_queries, k, steps=4):
    """Fabric traffic of THIS build's scan kernels on THIS box, measured outside the timed region: three
    rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hits / misses: separate passes, counters only — no trace
    domains) over a child process that runs `steps` synchronous pruned steps of the workload.  Units and
    corrections as MI355X_MICROARCH.md (HBM section) prescribes and profiles/fetch_calibration.json confirmed for
    this path's gathers: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 a 128-byte fabric read request is tallied as
    64 B, so read bytes = 2 x FETCH_SIZE; both counters sit on the fabric side of the L2 (Infinity-Cache hits
    included).  -> dict per step, or {"error": ...}."""
    import csv
    import glob
    import re
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    scan = re.compile(SCAN_KERNEL_RE)
    tot = {}       # counter -> total over the scan-kernel dispatches of the child
    by_kernel = {}  # kernel family -> counter -> total
    t0 = time.time()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]):
        tmp = tempfile.mkdtemp(prefix="tq_pmc_", dir="/tmp")
        cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", workload, "--docs", str(args.docs), "--terms", str(args.terms),
               "--queries", str(n_queries), "--k", str(k), "--steps", str(steps)]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            shutil.rmtree(tmp, ignore_errors=True)
            return {"error": "rocprofv3 pass timed out (%s)" % counters}
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            shutil.rmtree(tmp, ignore_errors=True)
            return {"error": "rocprofv3 pass failed (%s): rc %d %s" % (counters, r.returncode, (r.stderr or "")[-300:])}
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    m = scan.search(row["Kernel_Name"])
                    if not m:
                        continue
                    c, v = row["Counter_Name"], float(row["Counter_Value"])
                    tot[c] = tot.get(c, 0.0) + v
                    fam = by_kernel.setdefault(m.group(1), {})
                    fam[c] = fam.get(c, 0.0) + v
        shutil.rmtree(tmp, ignore_errors=True)
    if "FETCH_SIZE" not in tot:
        return {"error": "no scan-kernel dispatch in the counter files"}

    def fabric(d):
        rd = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024.0 / steps
        wr = d.get("WRITE_SIZE", 0.0) * 1024.0 / steps
        return rd, wr

    rd, wr = fabric(tot)
    hit, miss = tot.get("TCC_HIT_sum", 0.0), tot.get("TCC_MISS_sum", 0.0)
    out = {"traffic": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
           "read_requests": int(rd / 128.0), "write_requests": int(wr / 64.0),
           "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss > 0 else None,
           "l2_misses": int(miss / steps), "steps": steps, "seconds": round(time.time() - t0, 1),
           "by_kernel": {fam: {"traffic": int(sum(fabric(d))), "read_requests": int(fabric(d)[0] / 128.0)}
                         for fam, d in sorted(by_kernel.items())},
           "source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (three passes, counters only) around "
                     "`bench.py --pmc-child %s` in this run, %d synchronous pruned steps, scan-kernel dispatches only; "
                     "bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950: a 128-B fabric read request is tallied as 64 B) + "
                     "WRITE_SIZE KiB x 1024; requests: reads of 128 B, writes counted as 64 B each (uncalibrated)" %
                     (workload, steps)}
    return out


def stratified_sample(n_q, kernels_of, n_check):
    """Indices of the queries to check against the oracle: spread evenly over the batch, every scan-kernel family
    that ran gets at least min(32, its queries) of them (kernels_of: TQ_KERNEL_* bit per query, or None)."""
    base = set(int(x) for x in np.linspace(0, n_q - 1, min(n_check, n_q)))
    per = {}
    if kernels_of is not None:
        for fam in sorted(set(int(x) for x in kernels_of)):
            idx = np.nonzero(kernels_of == fam)[0]
            have = sum(1 for i in base if int(kernels_of[i]) == fam)
            want = min(32, len(idx))
            if have < want:
                extra = [int(idx[int(j)]) for j in np.linspace(0, len(idx) - 1, want)]
                base.update(extra)
            per[fam] = sum(1 for i in base if int(kernels_of[i]) == fam)
    return sorted(base), per


def spot_check(O, cl, segs, first_ord, gstats, workload, queries, k, final, n_check=64, kernels_of=None):
    """A sample of the batch against the oracle, at the run's own size: every rank runs the oracle's
    exhaustive executor on each of its local segments with the index-wide Bm25Weights, the hits
    are gathered over the control plane (gloo), merged by the oracle's merge_top_k (score desc,
    segment_ord asc, doc asc) and compared with the merged device result: doc addresses equal,
    scores within 1e-5 relative (BASELINE.json).  Returns the number of queries checked."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    n_q = len(queries)
    many = len(segs) > 1 or cl.world > 1
    sample, per_family = stratified_sample(n_q, kernels_of, n_check)

    def local_hits(i):
        hits = []
        for j, seg in enumerate(segs):
            spec = oracle_spec(O, seg, workload, queries[i], k, gstats if many else None)
            out = (O.Hit * max(1, k))()
            n = O.lib().to_search_exhaustive(C.byref(seg.view), C.byref(spec.q), out)
            hits += [(float(out[x].score), first_ord + j, int(out[x].doc)) for x in range(n)]
        return hits

    with ThreadPoolExecutor(max_workers=max(1, min(16, usable_cpus()))) as pool:
        mine = list(pool.map(local_hits, sample))  # (ctypes calls release the GIL)
    everyone = cl.all_gather_object(mine)
    for si, i in enumerate(sample):
        want = O.merge_top_k([h for r in everyone for h in r[si]], 0, k)
        got = [(float(final[0][i, j]), int(final[1][i, j]), int(final[2][i, j]))
               for j in range(int(final[3][i]))]
        assert len(got) == len(want), (workload, i, got, want)
        for (gs, go, gd), (ws, wo, wd) in zip(got, want):
            assert (go, gd) == (wo, wd) and abs(gs - ws) <= 1e-5 * abs(ws), (workload, i, got, want)
    spot_check.last_by_family = per_family
    return len(sample)


def query_kernels(runner, n_q):
    """TQ_KERNEL_* bit of every query of the prepared batch in the pruned mode (one extra untimed step with the
    library's "record_query_kernels" option; the local segments plan alike: segment 0 speaks for them)."""
    runner.set_option("record_query_kernels", 1)
    runner.set_option("exhaustive", 0)
    runner.enqueue()
    runner.synchronize()
    kern = runner.dev.last_batch_query_kernels(n_q, 0)
    runner.set_option("record_query_kernels", 0)
    return kern


def physical_roofline(k_ms, pm, committed, launches=1):
    """roofline.achieved / frac / traffic from bytes that crossed the fabric: this run's inline PMC passes if they
    ran, else the committed PMC run of the same kernel sources (profiles/traffic.json), else nothing."""
    out = {"traffic": None, "achieved": None, "frac": None, "l2_hit_rate": None, "traffic_source": "none: no PMC run "
           "(--no-pmc-inline or rocprofv3 failed) and no committed entry for this tree's kernels"}
    if pm and pm.get("traffic"):
        out.update(traffic=int(pm["traffic"] * launches), l2_hit_rate=pm.get("l2_hit_rate"), traffic_source=pm["source"],
                   read_requests=int(pm["read_requests"] * launches), write_requests=int(pm["write_requests"] * launches),
                   traffic_by_kernel=pm.get("by_kernel"), pmc_seconds=pm.get("seconds"))
    elif committed and committed.get("traffic") is not None:
        out.update(traffic=int(committed["traffic"]), l2_hit_rate=committed.get("l2_hit_rate"),
                   traffic_source="profiles/traffic.json, commit %s (same kernel sources): %s" %
                                  (committed.get("traffic_from_commit"), committed.get("traffic_note")))
    if pm and pm.get("error"):
        out["pmc_inline_error"] = pm["error"]
    if out["traffic"] is not None and k_ms > 0:
        out["achieved"] = round(out["traffic"] / (k_ms * 1e-3) / 1e9, 1)
        out["frac"] = round(out["achieved"] / HBM_PEAK_GBS, 4)
    return out


ASHARE_COUNTERS = (("blocks_decoded", 256), ("block_lead_pairs", 32), ("scoring_stage_candidates", 64), ("docs_collected", 0),
                   ("payload_bytes", 0x100000), ("fieldnorm_bytes", 0x200000), ("docmat_words", 0x400000),
                   ("range_maxima_bytes", 0x800000))


def ashare_useful_bytes(runner, queries, kern, k, ashare_bit, reads_ashare):
    """Bytes the lanes of the shared-intersection launch CONSUME per batch, from its work counters (one untimed
    step per counter on the sub-batch that rode in it — the library's "debug" option; no counter changes a
    result): bit-packed payload + block records of the decoded leader blocks, one fieldnorm byte per decoded doc,
    8 B per doc-matrix word gathered, the range-maxima bytes, 9 B (bitmap word + tf byte) per scoring-stage
    candidate, 12 B (staging entry + threshold slot) per collected doc.  Task and lead records (16 / 64 B per
    task / lead) are not counted.  request_efficiency = useful bytes / (128 B x the launch's fabric read requests)."""
    sub = [q for q, kk in zip(queries, kern) if int(kk) == ashare_bit]
    if len(sub) < 64:
        return None
    runner.prepare(sub, k)
    runner.set_option("exhaustive", 0)
    c = {}
    for name, bit in ASHARE_COUNTERS:
        runner.set_option("debug", bit)
        runner.enqueue()
        runner.synchronize()
        c[name] = int(runner.batch_stats()["matches"])
    runner.set_option("debug", -1)
    useful = (c["payload_bytes"] + c["fieldnorm_bytes"] + 8 * c["docmat_words"] + c["range_maxima_bytes"]
              + 9 * c["scoring_stage_candidates"] + 12 * c["docs_collected"])
    out = {"queries": len(sub), "counters": c, "useful_bytes": int(useful)}
    if reads_ashare:
        out["fabric_read_requests"] = int(reads_ashare)
        out["request_efficiency"] = round(useful / (128.0 * reads_ashare), 4)
    return out


def stream_block(O, D, cl, torch, args, vocab, seg, n_batches, n_q, k):
    """tantivy's serving pattern: every batch is NEW — seeds s, s+1, ... of the headline stream — and
    Query::weight (BM25 statistics, term lookups, tq_term_prepare of terms not seen before, their bitmaps /
    range maxima / probe tables on first use: tqh_prepare_batch) runs INSIDE the timed region, on a segment no
    query has touched.  Only the Python-side struct building of the request is done ahead.  Steps are pipelined
    like the replayed-batch loop (Searcher::search_with_executor, searcher.rs:180-238)."""
    runner = D.ShardRunner([seg], cl.local_rank)
    runner.set_option("timing", 1)
    batches = [[(O.MODE_AND, q.tolist()) for q in O.zipf_queries(n_q, 2, vocab, seed=20260921 + 7919 * (i + 1))]
               for i in range(n_batches)]
    marsh = [runner.dev.marshal(b) for b in batches]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.prepare(batches[0], k, marsh[0])  # cold: every term of the batch is prepared here
    t_prep0 = time.perf_counter() - t0
    runner.enqueue()
    runner.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    first = runner.results()
    prep_s, t1 = 0.0, time.perf_counter()
    n_warm = min(3, n_batches - 2)
    loop_s = [0.0, 0.0, 0.0]
    # Query::weight of batch i + 1 on a second host thread while this one plans and enqueues batch i (what a server's
    # request threads do); --stream-serial: one thread does both in turn
    overlap = not getattr(args, "stream_serial", False)
    if overlap:
        runner.prepare_next(batches[1], k, marsh[1])
    for i in range(1, n_batches):
        if i == 1 + n_warm:  # steady state from here: the first batches still meet new terms
            runner.synchronize()
            runner.batch_stats()
            prep_s, t1 = 0.0, time.perf_counter()
        ta = time.perf_counter()
        if overlap:
            prep_s += runner.commit_next()
            tb = time.perf_counter()
            if i + 1 < n_batches:
                runner.prepare_next(batches[i + 1], k, marsh[i + 1])
        else:
            tp = tb = time.perf_counter()
            runner.prepare(batches[i], k, marsh[i])
            prep_s += time.perf_counter() - tp
        tc = time.perf_counter()
        runner.enqueue()
        td = time.perf_counter()
        if i == 1 + n_warm:
            loop_s = [0.0, 0.0, 0.0]
        if i >= 1 + n_warm:
            loop_s[0] += tb - ta
            loop_s[1] += tc - tb
            loop_s[2] += td - tc
    ta = time.perf_counter()
    runner.synchronize()
    wall = time.perf_counter() - t1
    loop_s.append(time.perf_counter() - ta)
    timed = n_batches - 1 - n_warm
    st = runner.batch_stats()
    # second pass over the same batches: every term now has its tables (what the stream costs once a segment has been
    # served for a while — Query::weight, planning and execution of batches that still differ from one another)
    t2 = time.perf_counter()
    if overlap:
        runner.prepare_next(batches[1], k, marsh[1])
    for i in range(1, n_batches):
        if overlap:
            runner.commit_next()
            if i + 1 < n_batches:
                runner.prepare_next(batches[i + 1], k, marsh[i + 1])
        else:
            runner.prepare(batches[i], k, marsh[i])
        runner.enqueue()
    runner.synchronize()
    wall2 = time.perf_counter() - t2
    st2 = runner.batch_stats()
    # the last batch against the oracle (a stream that returned wrong rows fast would be worthless)
    checked = spot_check(O, cl, [seg], cl.rank, None, "and2", batches[-1], k, runner.results(), 256)
    seg_stats = runner.dev.segment_stats(0)
    runner.close()
    del first
    return {"terms": vocab, "batches": n_batches, "queries_per_batch": n_q, "k": k,
            "cold_first_batch_ms": round(cold_ms, 2), "cold_prepare_ms": round(t_prep0 * 1e3, 2),
            "steady_batches_timed": timed, "steady_qps": round(n_q * timed / wall, 1),
            "steady_ms_per_batch": round(wall / timed * 1e3, 3),
            "prepare_ms_per_batch": round(prep_s / timed * 1e3, 3),
            "prepare_share": round(prep_s / wall, 3),
            "loop_ms_per_batch": {"wait_for_weights": round(loop_s[0] / timed * 1e3, 3), "kick_next_weights": round(loop_s[1] / timed * 1e3, 3),
                                  "plan_and_enqueue": round(loop_s[2] / timed * 1e3, 3), "final_wait": round(loop_s[3] / timed * 1e3, 3)},
            "prepare_thread": "second host thread, one batch ahead" if overlap else "the enqueueing thread",
            "second_pass_qps": round(n_q * (n_batches - 1) / wall2, 1),
            "second_pass_ms_per_batch": round(wall2 / (n_batches - 1) * 1e3, 3),
            "second_pass_kernel_ms": round(st2["kernel_ms"], 4), "second_pass_host_plan_ms": round(st2["host_plan_ms"], 3),

            "kernel_ms_avg": round(st["kernel_ms"], 4), "host_plan_ms": round(st["host_plan_ms"], 3),
            "kernels": " + ".join(st.get("kernels") or []),
            "derived_bytes": seg_stats["derived_bytes"], "tantivy_bytes": seg_stats["tantivy_bytes"],
            "parity_checked_queries": checked}


def resident_bytes(runner):
    """Bytes the local segments keep in HBM, by kind, from the library (tq_segment_get_stats)."""
    tot = {}
    for s in range(runner.n_local):
        for key, v in runner.dev.segment_stats(s).items():
            if key == "device_scratch_bytes":  # one set per device, shared by its segments
                tot[key] = max(tot.get(key, 0), v)
            elif key.endswith("_bytes"):
                tot[key] = tot.get(key, 0) + v
    return tot


def main():
    args = parse_args()
    if args.pmc_child:
        raise SystemExit(pmc_child(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch_ranks(args))
    cl = Cluster()
    # (the host planner runs on the calling thread: TQ_PLAN_THREADS defaults to 1 — helper threads
    # were no faster on any workload here and a descheduled helper stalled a step for tens of ms)
    if cl.world != args.gpus and cl.world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, cl.world))
    if args.selftest_launcher:
        raise SystemExit(selftest_launcher(cl.rank, cl.world))
    import torch

    torch.set_num_threads(4)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if torch.cuda.device_count() <= cl.local_rank:
        raise SystemExit("rank %d has no GPU (%d visible)" % (cl.local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(cl.local_rank)
    cl.init(torch)

    from tantivy_amd import build as product_build

    if not os.path.exists(product_build.LIB):  # normally prebuilt (__graft_entry__.build())
        if cl.rank == 0:
            product_build.build()
        cl.barrier()
    from oracle import oracle as O  # workload generator + cpu baseline + spot check only
    from tantivy_amd import distributed as D

    rank, world = cl.rank, cl.world
    pruned_mode = not args.exhaustive
    n_main = args.queries or DEFAULT_QUERIES[args.workload]

    def stats_of(seg):
        return (seg.max_doc, seg.total_num_tokens, [t.doc_freq for t in seg.terms])

    # ---------------------------------------------------------------- main workload (weak)
    with_pos = args.workload in PHRASE_WORKLOADS
    t0 = time.time()
    if args.terms != 256 or args.segments != 1:
        args.no_side = True
    S_main = max(1, args.segments)
    if S_main > 1:
        args.no_cpu_baseline = True
        args.latency_queries = 0
    main_segs = [O.synth_segment(args.docs, n_terms=args.terms, segment_ord=rank * S_main + j,
                                 with_positions=with_pos, phrase_terms=PHRASE_TERMS) for j in range(S_main)]
    seg = main_segs[0]
    t_gen = time.time() - t0
    all_stats = cl.all_gather_object([stats_of(x) for x in main_segs])
    runner = D.ShardRunner(main_segs, cl.local_rank, rank, world,
                           [st for r, lst in enumerate(all_stats) if r != rank for st in lst])
    cl.make_comm(runner.dev.ctx)
    runner.comm, runner.torch_group = cl.comm, cl.torch_group
    runner.set_option("timing", 1)
    for name in ("dense_ratio", "dense_budget_x", "probe_budget_x", "docmat", "docsig", "device_prepare", "or_windows"):  # experiments: TQ_OPT_dense_ratio=...
        if os.environ.get("TQ_OPT_" + name):
            runner.set_option(name, int(os.environ["TQ_OPT_" + name]))
    if os.environ.get("BENCH_PREWARM"):  # experiment: prepare another workload's terms first (term handles,
        pq, _ = build_queries(O, os.environ["BENCH_PREWARM"], 10_000, None, args.terms)  # doc-matrix columns)
        runner.dev.prepare(pq)
    queries, k = build_queries(O, args.workload, n_main, args.k, args.terms)
    n_q = len(queries)
    n_distinct = len({(q[0], tuple(sorted(q[1]))) for q in queries})
    m = measure(cl, runner, torch, queries, k, args.steps, args.warmup, args.exhaustive)
    if not m["mode_parity"]:
        raise SystemExit("pruned and exhaustive results differ on %d queries" % m["n_diff"])

    # ---- single-query latency (p50), outside the timed region
    lat = []
    curve = None
    if rank == 0 and world == 1 and args.latency_queries > 0:
        for i in range(min(args.latency_queries, n_q)):
            runner.dev.prepare([queries[i]])
            t1 = time.perf_counter()
            runner.dev.search_prepared(k)
            lat.append(time.perf_counter() - t1)
        curve = latency_curve(runner.dev, queries, k)

    gstats_main = global_stats(all_stats)
    runner.prepare(queries, k)  # (the latency legs prepared other batches)
    kern_main = query_kernels(runner, n_q)  # which kernel family ran which query: the oracle sample covers every family
    parity_checked = spot_check(O, cl, main_segs, rank * S_main, gstats_main, args.workload, queries, k,
                                m["final"], args.check_queries, kern_main)
    parity_by_family = {tantivy_amd_kernel_name(b): c for b, c in spot_check.last_by_family.items()}
    main_resident = resident_bytes(runner)
    main_exchange_ms = runner.exchange_ms()
    # ---- fabric traffic of this run's kernels (rocprofv3 passes around a child process) and what the lanes consume
    # (the rocprofv3 passes themselves run after every timed leg of the run — three child processes per workload
    # keep the host's CPU quota busy for ten seconds, and a host-bound step measured right after them pays for it)
    pmc_on = world == 1 and S_main == 1 and not args.no_pmc_inline
    pmc_jobs = [("main", args.workload, n_q, k)] if (pmc_on and rank == 0 and pruned_mode) else []
    useful = None
    if world == 1 and S_main == 1 and pruned_mode and args.workload in ("and2", "and2_distinct"):
        from tantivy_amd import binding as TB

        useful = ashare_useful_bytes(runner, queries, kern_main, k, TB.KERNEL_ASHARE, None)
    # The CPU baselines run AFTER every GPU measurement of the run: 10 s of all granted host cores
    # exhaust the cgroup's CPU quota, and the host planner of the next GPU measurement pays for it
    # (one run had the or5 step at 8.1 ms right after the baseline, 5.2 ms without).
    cpu_jobs = []  # (key, segment, workload, queries, k, seconds, sweep)
    if world == 1 and not args.no_cpu_baseline:
        cpu_jobs.append(("main", [seg], args.workload, queries, k, args.cpu_seconds, True, None))

    # ---------------------------------------------------------------- other BASELINE configs (N=1)
    side = {}
    pos_seg = None
    pm_mixed = {}  # the mixed stream's inline PMC figures (config 5 at N = 1 runs the same stream on 8 segments)
    if world == 1 and not args.no_side:
        for wl in os.environ.get("BENCH_SIDE_ORDER", "and2_distinct,or5,phrase3,phrase3_adj,mixed,bool").split(","):
            if wl == args.workload:
                continue
            # Every side workload gets a DeviceIndex of its own (the same segment bytes uploaded again):
            # its terms are prepared in ITS queries' order, as in a `--workload <wl> --no-side` run —
            # the runs the profiles under profiles/ were taken from.  (On the main workload's index
            # the or5 kernel measured 2.75 ms instead of 2.36: term handles, doc-matrix columns and
            # table addresses follow the order in which the first workload prepared its terms.)
            s_seg = seg
            if (wl in PHRASE_WORKLOADS) != with_pos:
                if wl in PHRASE_WORKLOADS and pos_seg is not None:
                    s_seg = pos_seg  # (the segment the other phrase stream built)
                else:
                    s_seg = O.synth_segment(args.docs, n_terms=256, segment_ord=rank,
                                            with_positions=wl in PHRASE_WORKLOADS, phrase_terms=PHRASE_TERMS)
                if wl in PHRASE_WORKLOADS:
                    pos_seg = s_seg
            s_runner = D.ShardRunner([s_seg], cl.local_rank)
            s_runner.set_option("timing", 1)
            qs, kk = build_queries(O, wl, DEFAULT_QUERIES[wl], None)
            if os.environ.get("BENCH_SIDE_SLEEP"):  # experiment: let the GPU idle before a side workload
                time.sleep(float(os.environ["BENCH_SIDE_SLEEP"]))
            sm = measure(cl, s_runner, torch, qs, kk, args.side_steps, 3)
            if not sm["mode_parity"]:
                raise SystemExit("%s: pruned and exhaustive results differ on %d queries" % (wl, sm["n_diff"]))
            s_kern = query_kernels(s_runner, len(qs))
            checked = spot_check(O, cl, [s_seg], rank, None, wl, qs, kk, sm["final"], args.check_queries, s_kern)
            s_by_family = {tantivy_amd_kernel_name(b): c for b, c in spot_check.last_by_family.items()}
            k_ms = sm["stats"]["kernel_ms"]
            ach, fr = frac_of(sm["algo_bytes_full"], k_ms)
            # this run's own PMC passes for every workload (about 10 s each): roofline_frac = bytes that crossed the
            # fabric / kernel time / peak; the SURVEY 8d figure (algorithmic_frac) exceeds 1 where a launch shares lists
            if pmc_on:
                pmc_jobs.append((wl, wl, len(qs), kk))
            phys = physical_roofline(k_ms, None, None)  # (filled in after the PMC passes, at the end of the run)
            ach_e, fr_e = frac_of(sm["algo_bytes_full"], sm["exh_stats"]["kernel_ms"])
            side[wl] = {
                "config": "%s: %d queries/batch, k=%d, same %dM-doc segment%s" %
                          (wl, len(qs), kk, args.docs // 1_000_000,
                           " (with positions)" if wl in PHRASE_WORKLOADS else ""),
                "qps": round(len(qs) * args.side_steps / sm["elapsed"], 1),
                "ms_per_step": round(sm["elapsed"] / args.side_steps * 1e3, 3),
                "kernel_ms_avg": round(k_ms, 4),
                "roofline_achieved_GBps": phys["achieved"], "roofline_frac": phys["frac"],
                "traffic": phys["traffic"], "l2_hit_rate": phys["l2_hit_rate"], "traffic_source": phys["traffic_source"],
                "algorithmic_GBps": ach, "algorithmic_frac": fr,
                "exhaustive_kernel_ms": round(sm["exh_stats"]["kernel_ms"], 4),
                "exhaustive_algorithmic_frac": fr_e,
                "algorithmic_bytes_per_launch": int(sm["algo_bytes_full"]),
                "docs_scored_per_launch": int(sm["stats"]["matches"]),
                "host_plan_ms": round(sm["stats"]["host_plan_ms"], 3),
                "kernels": " + ".join(sm["stats"].get("kernels") or []),
                "distinct_queries": len({(q[0], tuple(sorted(q[1]))) + tuple(map(str, q[2:])) for q in qs}),
                "batch_unique_bytes": int(sm["stats"].get("unique_bytes", 0)),
                "pruned_equals_exhaustive": True, "parity_checked_queries": checked,
                "parity_checked_by_kernel": s_by_family,
            }
            if not args.no_cpu_baseline:
                cpu_jobs.append((wl, [s_seg], wl, qs, kk, max(2.0, args.cpu_seconds / 3), False, None))
            if s_runner is not runner:
                s_runner.close()
    runner.close()
    if cl.comm is not None:
        cl.comm.close()
        cl.comm = None

    # ---------------------------------------------------------------- config 5: 8 segments, strong
    strong = None
    if not args.no_side and N_SEGMENTS_STRONG % world == 0:
        s_local = N_SEGMENTS_STRONG // world
        ords = list(range(rank * s_local, (rank + 1) * s_local))
        t0 = time.time()
        segs = [O.synth_segment(args.docs, n_terms=256, segment_ord=o) for o in ords]
        t_gen8 = time.time() - t0
        mine = [stats_of(s) for s in segs]
        everyone = cl.all_gather_object(mine)
        remote = [st for r, lst in enumerate(everyone) if r != rank for st in lst]
        srun = D.ShardRunner(segs, cl.local_rank, rank, world, remote)
        cl.make_comm(srun.dev.ctx)
        srun.comm, srun.torch_group = cl.comm, cl.torch_group
        srun.set_option("timing", 1)
        qs, kk = build_queries(O, "mixed", DEFAULT_QUERIES["mixed"], None)
        strong_steps = max(3, args.side_steps // 4)
        sm = measure(cl, srun, torch, qs, kk, strong_steps, 1)
        if not sm["mode_parity"]:
            raise SystemExit("strong: pruned and exhaustive results differ on %d queries" % sm["n_diff"])
        k_ms = sm["stats"]["kernel_ms"]
        ach, fr = frac_of(sm["algo_bytes_full"], k_ms)
        gstats8 = global_stats(everyone)
        kern8 = query_kernels(srun, len(qs))
        checked8 = spot_check(O, cl, segs, ords[0], gstats8, "mixed", qs, kk, sm["final"], args.check_queries, kern8)
        by_family8 = {tantivy_amd_kernel_name(b): c for b, c in spot_check.last_by_family.items()}
        # fabric bytes per step and GPU: the mixed stream's figure of this run (one segment, inline PMC passes) times
        # the local segments — the per-segment traffic does not depend on how many segments share the GPU
        # (profiles/r04_and2_s8_pmc.md: same FETCH per segment with 8 segments resident)
        phys8 = physical_roofline(k_ms, None, None)  # (filled in after the PMC passes)
        res8 = resident_bytes(srun)
        strong = {
            "config": "BASELINE configs[4]: %d x %dM-doc segments (%dM docs), mixed 50%% 2-term AND / "
                      "50%% 5-term OR stream, %d queries/batch, k=%d, global BM25 statistics; %d "
                      "segment(s) per GPU, all-gather of the per-segment top-k + merge_top_k" %
                      (N_SEGMENTS_STRONG, args.docs // 1_000_000,
                       N_SEGMENTS_STRONG * args.docs // 1_000_000, len(qs), kk, s_local),
            "scaling": "strong", "unit": "queries/s over the whole %d-segment index" % N_SEGMENTS_STRONG,
            "n_gpus": world, "segments_per_gpu": s_local,
            "qps": round(len(qs) * strong_steps / sm["elapsed"], 1),
            "ms_per_step": round(sm["elapsed"] / strong_steps * 1e3, 3),
            "kernel_ms_per_gpu": round(k_ms, 4),
            "roofline_achieved_GBps_per_gpu": phys8["achieved"], "roofline_frac_per_gpu": phys8["frac"],
            "traffic_per_step_per_gpu": phys8["traffic"], "traffic_source": phys8["traffic_source"],
            "algorithmic_GBps_per_gpu": ach, "algorithmic_frac_per_gpu": fr,
            "algorithmic_bytes_per_step_per_gpu": int(sm["algo_bytes_full"]),
            "resident_bytes_per_gpu": res8,
            "hbm_resident_note": "tantivy's bytes %.3f GB + derived side tables %.3f GB resident per GPU "
                                 "(tq_segment_get_stats): %s the 256 MB Infinity Cache" %
                                 (res8.get("tantivy_bytes", 0) / 1e9, res8.get("derived_bytes", 0) / 1e9,
                                  "beyond" if res8.get("tantivy_bytes", 0) + res8.get("derived_bytes", 0) > 256e6
                                  else "within"),
            "host_plan_ms_per_gpu": round(sm["stats"]["host_plan_ms"], 3),
            "exchange_ms": round(srun.exchange_ms(), 4),
            "time_note": "per step and GPU: host_plan_ms = host time inside collect_segment over the "
                         "local segments (validate + plan + stage + enqueue; overlaps the previous step's "
                         "kernels while it stays below them), kernel_ms = scan kernels (HIP events), "
                         "exchange_ms = all-gather + merge_top_k (stream events)",
            "plan_threads": int(os.environ.get("TQ_PLAN_THREADS", "1")),
            "exchange": cl.exchange_note, "pruned_equals_exhaustive": True,
            "parity_checked_queries": checked8, "parity_checked_by_kernel": by_family8,
            "parity_note": "oracle (exhaustive executor per segment, global Bm25Weights) -> gathered over "
                           "the control plane -> oracle merge_top_k, against the merged device result",
            "index_build_s": round(t_gen8, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            cpu_jobs.append(("strong", segs, "mixed", qs, kk, args.cpu_seconds, False, gstats8))
        srun.close()
        if cl.comm is not None:
            cl.comm.close()

    if rank != 0:
        cl.barrier()
        if cl.dist is not None:
            cl.dist.destroy_process_group()
        return

    # the stream block: different batches, Query::weight inside the timed region (rank 0, one GPU)
    stream = None
    if world == 1 and S_main == 1 and not args.no_stream and args.workload == "and2":
        stream = {"note": "and2 stream, %d NEW batches per vocabulary (seeds differ), one fresh 10M-doc segment each: "
                          "tqh_prepare_batch (Query::weight: BM25 statistics, term lookups, tq_term_prepare and first-use "
                          "tables) inside the timed region, on a second host thread one batch ahead of the thread that plans and "
                          "enqueues (prepare_thread; --stream-serial: one thread does both), steps pipelined; steady_* = after the first %d batches; "
                          "second_pass_* = the same batches once more, when every term they name has its tables (a segment that has "
                          "been served for a while: Query::weight, planning and execution still per batch); "
                          "replayed_qps = the headline loop's figure on the same vocabulary (one prepared batch replayed)" %
                          (args.stream_batches, 1 + min(3, args.stream_batches - 2)), "by_terms": {}}
        for vocab in [int(x) for x in args.stream_vocabs.split(",") if x]:
            t_s = time.time()
            v_seg = main_segs[0] if (vocab == args.terms and not with_pos) else O.synth_segment(args.docs, n_terms=vocab, segment_ord=rank)
            sb = stream_block(O, D, cl, torch, args, vocab, v_seg, args.stream_batches, n_q, k)
            sb["derived_x"] = round(sb["derived_bytes"] / max(1, sb["tantivy_bytes"]), 2)
            sb["seconds"] = round(time.time() - t_s, 1)
            stream["by_terms"][str(vocab)] = sb
            del v_seg
    # ---- this run's PMC passes, after everything that is timed: fabric bytes per workload -> roofline fields
    pm_main = None
    for key, wl, nn, kk in pmc_jobs:
        pm = pmc_inline(args, wl, nn, kk)
        if key == "main":
            pm_main = pm
            continue
        sw = side[key]
        ph = physical_roofline(sw["kernel_ms_avg"], pm, traffic_fields(load_traffic("%s_pruned_%d" % (wl, args.docs)), 1, sw["kernel_ms_avg"]))
        sw.update({"roofline_achieved_GBps": ph["achieved"], "roofline_frac": ph["frac"], "traffic": ph["traffic"],
                   "l2_hit_rate": ph["l2_hit_rate"], "traffic_source": ph["traffic_source"]})
        if key == "mixed" and strong is not None and pm.get("traffic"):
            # config 5 at this N: the mixed stream's figure of this run (one segment) times the local segments — the
            # per-segment traffic does not depend on how many segments share the GPU (profiles/r04_and2_s8_pmc.md)
            s_loc = strong["segments_per_gpu"]
            ph8 = physical_roofline(strong["kernel_ms_per_gpu"], pm, None, launches=s_loc)
            strong.update({"roofline_achieved_GBps_per_gpu": ph8["achieved"], "roofline_frac_per_gpu": ph8["frac"],
                           "traffic_per_step_per_gpu": ph8["traffic"],
                           "traffic_source": "%d x the mixed stream's fabric bytes per batch of this run (%s)" % (s_loc, ph8["traffic_source"])})
    if useful and pm_main and pm_main.get("by_kernel", {}).get("ashare_kernel"):
        reads_a = pm_main["by_kernel"]["ashare_kernel"]["read_requests"]
        useful["fabric_read_requests"] = int(reads_a)
        useful["request_efficiency"] = round(useful["useful_bytes"] / (128.0 * reads_a), 4) if reads_a else None
    cpu = None
    for key, c_segs, c_wl, c_qs, c_k, c_sec, c_sweep, c_gs in cpu_jobs:
        c = cpu_baseline(O, c_segs, c_wl, c_qs, c_k, c_sec, sweep=c_sweep, gstats=c_gs)
        if key == "main":
            cpu = c
        elif key == "strong":
            strong["cpu_baseline"] = c
        else:
            side[key]["cpu_baseline"] = {kk2: c[kk2] for kk2 in ("value", "unit", "cores", "kind",
                                                                 "qps_simd", "qps_scalar")}

    st = m["stats"]
    k_ms = st["kernel_ms"]
    algo_bytes = m["algo_bytes_full"]
    achieved, frac = frac_of(algo_bytes, k_ms)
    uniq_bytes = (st.get("unique_bytes", 0) + sum(x.max_doc for x in main_segs if x.fieldnorm is not None)
                  + 8 * k * n_q) * 1
    other_st = m["exh_stats"] if pruned_mode else m["prn_stats"]
    o_ach, o_frac = frac_of(algo_bytes, other_st["kernel_ms"])
    tkey = "%s_%s_%d" % (args.workload, "pruned" if pruned_mode else "exhaustive", args.docs)
    if args.terms != 256:
        tkey += "_t%d" % args.terms
    if S_main != 1:
        tkey += "_s%d" % S_main
    tf = traffic_fields(load_traffic(tkey), S_main, k_ms)  # per step: one scan launch per local segment
    phys = physical_roofline(k_ms, pm_main, tf)
    total_units = n_q * args.steps * world  # one unit = one query evaluated on one segment
    value = total_units / m["elapsed"]
    out = {
        "metric": "queries_per_sec_2term_AND_bm25_top10" if args.workload == "and2"
        else "queries_per_sec_" + args.workload,
        "value": round(value, 1),
        "value_distinct": (side.get("and2_distinct") or {}).get("qps"),  # the same stream without repeated queries (other_workloads.and2_distinct)
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32+f32",
        "data": "synthetic",
        "config": {
            "workload": "%s: %d queries/batch, k=%d, one %dM-doc Zipf segment per GPU (%d terms, "
                        "df_r=0.5N/r, WithFreqs%s); term ranks ~ Zipf(1)" %
                        (args.workload, n_q, k, args.docs // 1_000_000, args.terms,
                         "AndPositions" if with_pos else ""),
            "distinct_queries": n_distinct,
            "unit_note": "weak scaling: one unit = one query evaluated on one segment; at N GPUs every "
                         "query runs on N segments (N x %dM docs), the per-segment top-k are "
                         "all-gathered over RCCL and merged.  The same 8-segment index at every N "
                         "(strong scaling, config 5) is reported under strong_scaling" %
                         (args.docs // 1_000_000),
            "timed_region": "K x [collect_segment (plan + H2D of query descriptors + scan + merge "
                            "kernels) -> all-gather -> merge_top_k -> D2H], enqueued back to back "
                            "(host planning of step i+1 overlaps the GPU work of step i), one "
                            "synchronisation at each end",
            "mode": "block-max pruned (block_wand_intersection semantics)" if pruned_mode
                    else "exhaustive (every match scored)",
            "exchange": cl.exchange_note,
            "segments_per_gpu": S_main,
            "index_bytes": int(sum(x.idx_len for x in main_segs)),
            "index_build_s": round(t_gen, 2),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": phys["achieved"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": phys["frac"],
            "traffic": phys["traffic"],
            "traffic_source": phys["traffic_source"],
            "fabric_read_requests": phys.get("read_requests"),
            "fabric_write_requests": phys.get("write_requests"),
            "traffic_by_kernel": phys.get("traffic_by_kernel"),
            "l2_hit_rate": phys["l2_hit_rate"],
            "pmc_inline_error": phys.get("pmc_inline_error"),
            "pmc_inline_seconds": phys.get("pmc_seconds"),
            "shared_launch_useful_bytes": useful,
            "committed_pmc_run": {"traffic": tf["traffic"], "physical_frac": tf["physical_frac"],
                                  "from_commit": tf["traffic_from_commit"], "matches_this_build": tf["traffic_matches_this_build"]},
            "kernel": " + ".join(st.get("kernels") or [args.workload + " scan kernels"]),
            "kernel_ms_avg": round(k_ms, 4),
            "algorithmic_GBps": achieved,
            "algorithmic_frac": frac,
            "algorithmic_bytes_per_launch": int(algo_bytes),
            "batch_unique_bytes": int(uniq_bytes),
            "unique_frac": round(uniq_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
            "unique_note": "batch_unique_bytes = every DISTINCT posting list of the batch once (tq_batch_stats."
                           "unique_bytes) + the fieldnorm file once + 8k per query: what the launch needs from "
                           "HBM when the queries of a batch share what they read (the term-major launches "
                           "do); algorithmic_bytes_per_launch counts a list once per query that names it "
                           "(SURVEY.md 8d), so frac can exceed 1 for a batch whose queries share lists — "
                           "that is sharing, not skipped work: unique_frac is the floor",
            "docs_scored_per_launch": int(st["matches"]),
            "launch_tasks": int(st.get("chunks", 0)),
            "matches_per_launch": int(m["full_matches"]),
            "traffic_note": tf["traffic_note"],
            "host_plan_ms": round(st["host_plan_ms"], 3),
            "gpu_batch_ms": round(st["total_ms"], 4),  # the batch on its stream: zeroing + scan kernels + merges (HIP events)
            "exchange_ms": round(main_exchange_ms, 4),
            "resident_bytes": main_resident,
            "frac_note": "achieved / frac = bytes that crossed the L2's fabric side (traffic: this run's rocprofv3 "
                         "passes, see traffic_source; Infinity-Cache hits included) / scan-kernel time (HIP events on "
                         "the launch streams) / 8 TB/s: a fraction of the HBM peak, <= 1 by construction.  "
                         "algorithmic_* = SURVEY.md 8d bytes (postings ranges of every query + 1 B per match + 8k: what "
                         "one scan PER QUERY would read) / the same time: work-equivalent bandwidth — above the peak "
                         "when the queries of a batch share leader blocks and block-max pruning skips the rest; "
                         "shared_launch_useful_bytes = what the lanes of the shared launch consume (work counters), "
                         "request_efficiency = that / (128 B x its fabric read requests).  A launch that gets faster by "
                         "moving FEWER bytes lowers this fraction: the headline batch went from 5.34 GB in 0.92 ms (round 5: "
                         "frac 0.72) to 2.85 GB in 0.81 ms (round 6: frac 0.44) — same queries, same results, same "
                         "algorithmic bytes (algorithmic_frac 3.43 -> 3.92)",
        },
        "roofline_other_mode": {"mode": "exhaustive" if pruned_mode else "pruned", "achieved": o_ach,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": o_frac,
                                "kernel_ms_avg": round(other_st["kernel_ms"], 4)},
        "pruned_equals_exhaustive": bool(m["mode_parity"]),
        "cpu_baseline": cpu,
        "p50_latency_ms": round(float(np.median(lat)) * 1e3, 4) if lat else None,
        "p50_latency_note": "one query per call, one caller (the batch = 1 point of latency_curve); the "
                            "headline batch's own p50 is latency_curve.batch['10000'].p50_ms",
        "latency_curve": curve,
        "parity_checked_queries": parity_checked,
        "parity_checked_by_kernel": parity_by_family,
        "stream": stream,
        "other_workloads": side or None,
        "strong_scaling": strong,
    }
    print(json.dumps(out))
    sys.stdout.flush()
    cl.barrier()
    if cl.dist is not None:
        cl.dist.destroy_process_group()


if __name__ == "__main__":
    main()
