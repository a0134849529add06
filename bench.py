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
    ap.add_argument("--steps", type=int, default=20)
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
    ap.add_argument("--workload", default="and2", choices=["and2", "and2_distinct", "or5", "phrase3", "mixed", "bool"])
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
    ap.add_argument("--side-steps", type=int, default=10)
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU plumbing check of the rank launcher (gloo, no GPU work, no timing)")
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
                   "bool": 2_000}


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
    return O.QuerySpec(seg, q[1], w, q[0], k,
                       list(range(len(q[1]))) if q[0] == O.MODE_PHRASE else None)


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
    for b in (1, 16, 256, 4096, 10000):
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
    for nthreads in (1, 16, 64):
        n = min(len(queries), 400 if nthreads == 1 else nthreads * 150)
        dev.search_concurrent(queries[:min(n, 4 * nthreads)], k, nthreads)  # (threads + queue warm)
        dev.submit_stats(reset=True)
        _, _, _, _, lat_ms, wall_ms = dev.search_concurrent(queries[:n], k, nthreads)
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


def spot_check(O, cl, segs, first_ord, gstats, workload, queries, k, final, n_check=64):
    """A sample of the batch against the oracle, at the run's own size: every rank runs the oracle's
    exhaustive executor on each of its local segments with the index-wide Bm25Weights, the hits
    are gathered over the control plane (gloo), merged by the oracle's merge_top_k (score desc,
    segment_ord asc, doc asc) and compared with the merged device result: doc addresses equal,
    scores within 1e-5 relative (BASELINE.json).  Returns the number of queries checked."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    n_q = len(queries)
    many = len(segs) > 1 or cl.world > 1
    sample = sorted(set(int(x) for x in np.linspace(0, n_q - 1, min(n_check, n_q))))

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
    return len(sample)


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
    with_pos = args.workload == "phrase3"
    t0 = time.time()
    if args.terms != 256 or args.segments != 1:
        args.no_side = True
    S_main = max(1, args.segments)
    if S_main > 1:
        args.no_cpu_baseline = True
        args.latency_queries = 0
    main_segs = [O.synth_segment(args.docs, n_terms=args.terms, segment_ord=rank * S_main + j,
                                 with_positions=with_pos, phrase_terms=32) for j in range(S_main)]
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
    parity_checked = spot_check(O, cl, main_segs, rank * S_main, gstats_main, args.workload, queries, k,
                                m["final"], 64)
    main_resident = resident_bytes(runner)
    main_exchange_ms = runner.exchange_ms()
    # The CPU baselines run AFTER every GPU measurement of the run: 10 s of all granted host cores
    # exhaust the cgroup's CPU quota, and the host planner of the next GPU measurement pays for it
    # (one run had the or5 step at 8.1 ms right after the baseline, 5.2 ms without).
    cpu_jobs = []  # (key, segment, workload, queries, k, seconds, sweep)
    if world == 1 and not args.no_cpu_baseline:
        cpu_jobs.append(("main", [seg], args.workload, queries, k, args.cpu_seconds, True, None))

    # ---------------------------------------------------------------- other BASELINE configs (N=1)
    side = {}
    if world == 1 and not args.no_side:
        for wl in os.environ.get("BENCH_SIDE_ORDER", "and2_distinct,or5,phrase3,mixed,bool").split(","):
            if wl == args.workload:
                continue
            # Every side workload gets a DeviceIndex of its own (the same segment bytes uploaded again):
            # its terms are prepared in ITS queries' order, as in a `--workload <wl> --no-side` run —
            # the runs the profiles under profiles/ were taken from.  (On the main workload's index
            # the or5 kernel measured 2.75 ms instead of 2.36: term handles, doc-matrix columns and
            # table addresses follow the order in which the first workload prepared its terms.)
            s_seg = seg
            if (wl == "phrase3") != with_pos:
                s_seg = O.synth_segment(args.docs, n_terms=256, segment_ord=rank,
                                        with_positions=wl == "phrase3", phrase_terms=32)
            s_runner = D.ShardRunner([s_seg], cl.local_rank)
            s_runner.set_option("timing", 1)
            qs, kk = build_queries(O, wl, DEFAULT_QUERIES[wl], None)
            if os.environ.get("BENCH_SIDE_SLEEP"):  # experiment: let the GPU idle before a side workload
                time.sleep(float(os.environ["BENCH_SIDE_SLEEP"]))
            sm = measure(cl, s_runner, torch, qs, kk, args.side_steps, 2)
            if not sm["mode_parity"]:
                raise SystemExit("%s: pruned and exhaustive results differ on %d queries" % (wl, sm["n_diff"]))
            checked = spot_check(O, cl, [s_seg], rank, None, wl, qs, kk, sm["final"], 64)
            k_ms = sm["stats"]["kernel_ms"]
            ach, fr = frac_of(sm["algo_bytes_full"], k_ms)
            ach_e, fr_e = frac_of(sm["algo_bytes_full"], sm["exh_stats"]["kernel_ms"])
            side[wl] = {
                "config": "%s: %d queries/batch, k=%d, same %dM-doc segment%s" %
                          (wl, len(qs), kk, args.docs // 1_000_000,
                           " (with positions)" if wl == "phrase3" else ""),
                "qps": round(len(qs) * args.side_steps / sm["elapsed"], 1),
                "ms_per_step": round(sm["elapsed"] / args.side_steps * 1e3, 3),
                "kernel_ms_avg": round(k_ms, 4),
                "roofline_achieved_GBps": ach, "roofline_frac": fr,
                "exhaustive_kernel_ms": round(sm["exh_stats"]["kernel_ms"], 4),
                "exhaustive_roofline_frac": fr_e,
                "algorithmic_bytes_per_launch": int(sm["algo_bytes_full"]),
                "docs_scored_per_launch": int(sm["stats"]["matches"]),
                "host_plan_ms": round(sm["stats"]["host_plan_ms"], 3),
                "kernels": " + ".join(sm["stats"].get("kernels") or []),
                "distinct_queries": len({(q[0], tuple(sorted(q[1]))) + tuple(map(str, q[2:])) for q in qs}),
                "batch_unique_bytes": int(sm["stats"].get("unique_bytes", 0)),
                "pruned_equals_exhaustive": True, "parity_checked_queries": checked,
            }
            side[wl].update(traffic_fields(load_traffic("%s_pruned_%d" % (wl, args.docs)), 1, k_ms))
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
        strong_steps = max(3, args.side_steps // 2)
        sm = measure(cl, srun, torch, qs, kk, strong_steps, 1)
        if not sm["mode_parity"]:
            raise SystemExit("strong: pruned and exhaustive results differ on %d queries" % sm["n_diff"])
        k_ms = sm["stats"]["kernel_ms"]
        ach, fr = frac_of(sm["algo_bytes_full"], k_ms)
        gstats8 = global_stats(everyone)
        checked8 = spot_check(O, cl, segs, ords[0], gstats8, "mixed", qs, kk, sm["final"], 64)
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
            "roofline_achieved_GBps_per_gpu": ach, "roofline_frac_per_gpu": fr,
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
            "parity_checked_queries": checked8,
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
    total_units = n_q * args.steps * world  # one unit = one query evaluated on one segment
    value = total_units / m["elapsed"]
    out = {
        "metric": "queries_per_sec_2term_AND_bm25_top10" if args.workload == "and2"
        else "queries_per_sec_" + args.workload,
        "value": round(value, 1),
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
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": frac,
            "traffic": tf["traffic"],
            "physical_frac": tf["physical_frac"],
            "l2_hit_rate": tf["l2_hit_rate"],
            "traffic_from_commit": tf["traffic_from_commit"],
            "traffic_matches_this_build": tf["traffic_matches_this_build"],
            "kernel": " + ".join(st.get("kernels") or [args.workload + " scan kernels"]),
            "kernel_ms_avg": round(k_ms, 4),
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
            "matches_per_launch": int(m["full_matches"]),
            "traffic_note": tf["traffic_note"],
            "host_plan_ms": round(st["host_plan_ms"], 3),
            "exchange_ms": round(main_exchange_ms, 4),
            "resident_bytes": main_resident,
            "frac_note": "frac = algorithmic bytes (SURVEY.md §8d: postings ranges + 1 B per match + 8k; "
                         "what a full scan would read) / kernel time / 8 TB/s — the pruned kernel skips "
                         "most of them, so this is work-equivalent bandwidth, not achieved HBM bandwidth; "
                         "physical_frac = rocprofv3 fabric bytes (committed PMC run, calibrated per "
                         "DESIGN.md §3.0) / kernel time / 8 TB/s",
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
