"""Segments sharded over GPUs, one process per GPU: every rank holds a contiguous
run of the index's segments, all ranks run the same query batch with the same global BM25
statistics, and the per-segment top-k lists are exchanged with ONE all-gather (RCCL over xGMI
through the C ABI's tq_allgather_topk; gloo on CPU in the tests) before `merge_top_k`
(src/collector/sort_key_top_collector.rs:76-95) picks the global top-(offset+limit) by
(score desc, segment_ord asc, doc asc).  No other collective exists on this path: BM25
statistics are sums of per-segment counters known to the host before dispatch (bm25.rs:27-50).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import binding as B


def allgather_topk(scores, docs, counts, group=None):
    """torch.distributed version (gloo in the CPU tests; RCCL fallback when the C-ABI communicator
    is unavailable).  scores/docs: [rows, k] tensors, counts: [rows].  Returns the gathered
    [world, rows, k] / [world, rows] tensors (rank order == segment order)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n, k = scores.shape
    # one message per rank: scores | docs | counts packed as int32 words
    packed = torch.empty(n * (2 * k + 1), dtype=torch.int32, device=scores.device)
    packed[: n * k] = scores.contiguous().view(torch.int32).reshape(-1)
    packed[n * k: 2 * n * k] = docs.contiguous().view(torch.int32).reshape(-1)
    packed[2 * n * k:] = counts.contiguous().view(torch.int32).reshape(-1)
    out = torch.empty(world * packed.numel(), dtype=torch.int32, device=scores.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out.view(world, -1)
    g_scores = out[:, : n * k].contiguous().view(torch.float32).view(world, n, k)
    g_docs = out[:, n * k: 2 * n * k].contiguous().view(world, n, k)
    g_counts = out[:, 2 * n * k:].contiguous().view(world, n)
    return g_scores, g_docs, g_counts


def merge_gathered_host(g_scores, g_docs, g_counts, offset, limit):
    """merge_top_k on the host (tq_merge_topk) over gathered per-segment results."""
    sc = np.ascontiguousarray(g_scores.cpu().numpy(), np.float32)
    dc = np.ascontiguousarray(g_docs.cpu().numpy()).view(np.uint32)
    ct = np.ascontiguousarray(g_counts.cpu().numpy()).view(np.uint32)
    S, n, k = sc.shape
    out_s = np.zeros((n, limit), np.float32)
    out_o = np.zeros((n, limit), np.uint32)
    out_d = np.zeros((n, limit), np.uint32)
    out_c = np.zeros(n, np.uint32)
    B._check(B.lib().tq_merge_topk(B._f32(sc), B._u32(dc), B._u32(ct), S, n, k, offset, limit,
                                   B._f32(out_s), B._u32(out_o), B._u32(out_d), B._u32(out_c)))
    return out_s, out_o, out_d, out_c


def merge_gathered_device(ctx, device, g_scores, g_docs, g_counts, offset, limit, stream=None,
                          out=None):
    """merge_top_k on the device (tq_merge_topk_device); tensors stay on the GPU."""
    import torch

    S, n, k = g_scores.shape
    dev = g_scores.device
    if out is None:
        out = (torch.empty((n, limit), dtype=torch.float32, device=dev),
               torch.empty((n, limit), dtype=torch.int32, device=dev),
               torch.empty((n, limit), dtype=torch.int32, device=dev),
               torch.empty(n, dtype=torch.int32, device=dev))
    out_s, out_o, out_d, out_c = out
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    B._check(B.lib().tq_merge_topk_device(
        ctx, int(device), g_scores.data_ptr(), g_docs.data_ptr(), g_counts.data_ptr(), None, S, n, k,
        int(offset), int(limit), out_s.data_ptr(), out_o.data_ptr(), out_d.data_ptr(),
        out_c.data_ptr(), C.c_void_p(st)))
    return out_s, out_o, out_d, out_c


class Comm:
    """RCCL communicator behind the C ABI (tq_comm_*): what a Rust host would hold.  `exchange`
    sends rank 0's 128-byte id to every rank (any control-plane channel; here the caller's)."""

    def __init__(self, ctx, device, rank, world, exchange):
        L = B.lib()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            B._check(L.tq_comm_unique_id(ident))
        raw = exchange(bytes(ident) if rank == 0 else None)
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        self._c = C.c_void_p()
        B._check(L.tq_comm_init(ctx, int(device), ident, int(rank), int(world), C.byref(self._c)))
        self.rank, self.world = rank, world
        lib_name = C.c_char_p()
        B._check(L.tq_comm_info(self._c, None, None, C.byref(lib_name)))
        self.library = (lib_name.value or b"").decode()

    def allgather_topk(self, scores, docs, counts, out, stream):
        """scores/docs [rows, k], counts [rows] -> out = ([world, rows, k] x2, [world, rows])."""
        rows, k = scores.shape
        B._check(B.lib().tq_allgather_topk(
            self._c, scores.data_ptr(), docs.data_ptr(), counts.data_ptr(), rows, k,
            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), C.c_void_p(stream)))

    def close(self):
        if self._c:
            B.lib().tq_comm_free(self._c)
            self._c = C.c_void_p()


def exchange_topk(local, n, k, n_local, world, comm=None, torch_group=None, gathered=None, stream=None,
                  force=False):
    """The path's one exchange step: every rank's [S * n, k] slabs (S local segments, n queries) ->
    the per-segment results of the whole index as [world * S, n, k] / [world * S, n] views, rank order
    == segment order (rank r holds segments r * S .. r * S + S - 1).  `comm`: an object with
    allgather_topk(scores, docs, counts, out, stream) filling out = ([world, S * n, k] x 2,
    [world, S * n]) — the C-ABI communicator (tq_allgather_topk: three gathers in one grouped RCCL
    launch); else torch.distributed's group (one packed gather).  world == 1: no exchange."""
    sc, dc, ct = local
    S, W = n_local, world
    if W > 1 or force:
        if comm is not None:
            comm.allgather_topk(sc, dc, ct, gathered, stream)
            g = gathered
        else:  # RCCL through torch.distributed (same wire, torch's communicator)
            g = allgather_topk(sc, dc, ct, group=torch_group)
    else:
        g = (sc.unsqueeze(0), dc.unsqueeze(0), ct.unsqueeze(0))
    # [world][S*n][k] == [world*S segments][n][k]
    return g[0].reshape(W * S, n, k), g[1].reshape(W * S, n, k), g[2].reshape(W * S, n)


class ShardRunner:
    """One rank's share of a sharded index: `segments` (its contiguous run, in global segment
    order) resident on `device`, the global statistics of the others added as remote statistics.
    enqueue() = for every local segment collect_segment (tq_search_batch_device) into its slab,
    one all-gather of the slabs, merge_top_k on the device, asynchronous copies to pinned host
    memory — all on one HIP stream, nothing waits for the GPU."""

    def __init__(self, segments, device, rank=0, world=1, remote_stats=(), comm=None,
                 torch_group=None, force_exchange=False):
        import torch

        from . import DeviceIndex

        self.torch = torch
        self.dev = DeviceIndex(segments, devices=[device] * max(1, len(segments)))
        self.n_local = len(segments)
        self.device, self.rank, self.world = device, rank, world
        for st in remote_stats:
            self.dev.add_remote_stats(*st)
        self.comm, self.torch_group = comm, torch_group
        self.force_exchange = force_exchange  # tests: run the all-gather even with one rank
        self.stream_obj = torch.cuda.Stream(device=device)
        self.stream = self.stream_obj.cuda_stream
        # (timing events are created once: two hipEventCreate per step inside a pipelined loop now and then cost
        # milliseconds when the runtime grew its pools)
        self._ev_ring = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(16)]
        self._ev_at = 0
        self._trace = bool(os.environ.get("TQ_RUNNER_TRACE"))
        self.n = self.k = 0
        self._ex_events = []  # (start, end) torch events around exchange + merge, last 16 steps
        self._agree("local segments per rank", self.n_local)

    def _agree(self, what, value):
        """The equal-count all-gather and the [world * S, n, k] reshape need every rank to hold the
        same number of local segments and to run the same (n, k): mismatched ranks would hang the
        collective or merge garbage, so the control plane (torch.distributed's default group) checks
        it where one exists."""
        if self.world <= 1:
            return
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, value)
        if any(v != value for v in seen):
            raise ValueError("ShardRunner: ranks disagree on %s: %r" % (what, seen))

    def _any_rank(self, flag):
        """True on every rank iff `flag` is true on at least one (control plane; the local flag where
        there is none)."""
        if self.world <= 1:
            return bool(flag)
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return bool(flag)
        dev = self.torch.device("cuda", self.device) if dist.get_backend() == "nccl" else "cpu"
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def set_option(self, name, value):
        self.dev.set_option(name, value)

    def prepare(self, queries, k, marshalled=None):
        """Query::weight for the batch; `marshalled` = DeviceIndex.marshal(queries) done earlier (stream benches:
        the Python-side struct building stays out of the timed region, Query::weight stays in)."""
        torch = self.torch
        if marshalled is not None:
            self.dev.prepare_marshalled(marshalled)
        else:
            self.dev.prepare(queries)
        n, S, W = len(queries), self.n_local, self.world
        changed = (n, k) != (self.n, self.k)
        # Every rank enters the agreement whenever ANY rank's shape changed (one small all-reduce per
        # prepare, not per step): a rank that changed alone would otherwise block in the all-gather of
        # _agree while the others go on to the RCCL gather with other sizes.
        if self._any_rank(changed):
            self._agree("(queries per batch, k)", (n, k))
        if not changed:
            return
        self.n, self.k = n, k
        cuda = torch.device("cuda", self.device)
        self.local = (torch.empty((S * n, k), dtype=torch.float32, device=cuda),
                      torch.empty((S * n, k), dtype=torch.int32, device=cuda),
                      torch.empty(S * n, dtype=torch.int32, device=cuda))
        if W > 1 or self.force_exchange:
            self.gathered = (torch.empty((W, S * n, k), dtype=torch.float32, device=cuda),
                             torch.empty((W, S * n, k), dtype=torch.int32, device=cuda),
                             torch.empty((W, S * n), dtype=torch.int32, device=cuda))
        # merged rows: scores | segment ordinals | docs | counts as views of ONE buffer, so that a step's
        # results leave the device in one copy (four copies cost 75 us of a 1.7 ms step in stream order)
        self._merged_all = torch.empty(n * (3 * k + 1), dtype=torch.int32, device=cuda)
        self._host_all = torch.empty(n * (3 * k + 1), dtype=torch.int32).pin_memory()

        def views(buf):
            return (buf[0:n * k].view(torch.float32).view(n, k), buf[n * k:2 * n * k].view(n, k),
                    buf[2 * n * k:3 * n * k].view(n, k), buf[3 * n * k:3 * n * k + n])

        # merge_top_k writes its rows STRAIGHT into the pinned host buffer (hipHostMalloc memory is mapped into the
        # device's address space): no device-to-host copy operation at all.  An asynchronous copy — torch's copy_, then
        # hipMemcpyAsync through the C ABI — blocked the calling thread for 7 ms now and then (one to three of ten
        # 20-step runs lost a third of their time to ONE such call; tools/r5_hiccup.sh, TQ_RUNNER_TRACE=1).
        # TQ_RUNNER_D2H_COPY=1: merge into device memory, then tq_copy_to_host_async.
        self._zero_copy = os.environ.get("TQ_RUNNER_D2H_COPY", "0") != "1"
        self.host = list(views(self._host_all))
        self.merged = tuple(self.host) if self._zero_copy else views(self._merged_all)

    def prepare_next(self, queries, k, marshalled):
        """Query::weight of the NEXT batch on a helper thread while this thread enqueues the current one (same number
        of queries and k as the current batch: nothing to agree on, no buffer changes); commit_next() switches to it."""
        if (len(queries), k) != (self.n, self.k):
            raise ValueError("prepare_next: the next batch must have the shape of the current one")
        self.dev.prepare_next_async(marshalled)

    def commit_next(self):
        return self.dev.commit_next()

    def enqueue(self):
        torch = self.torch
        n, k, S, W = self.n, self.k, self.n_local, self.world
        sc, dc, ct = self.local
        tr = self._trace  # (TQ_RUNNER_TRACE=1: host time of every part of a step that took more than 3 ms)
        t0 = time.perf_counter() if tr else 0.0
        with torch.cuda.stream(self.stream_obj):
            t0b = time.perf_counter() if tr else 0.0
            for s in range(S):
                self.dev.collect_segment_prepared_device(
                    s, k, sc[s * n:(s + 1) * n], dc[s * n:(s + 1) * n], ct[s * n:(s + 1) * n],
                    self.stream)
            t1 = time.perf_counter() if tr else 0.0
            ev = self._ev_ring[self._ev_at % 16]
            self._ev_at += 1
            ev[0].record(self.stream_obj)
            t2 = time.perf_counter() if tr else 0.0
            g = exchange_topk((sc, dc, ct), n, k, S, W, self.comm, self.torch_group,
                              getattr(self, "gathered", None), self.stream, self.force_exchange)
            merge_gathered_device(self.dev.ctx, self.device, g[0], g[1], g[2], 0, k, self.stream,
                                  out=self.merged)
            t3 = time.perf_counter() if tr else 0.0
            ev[1].record(self.stream_obj)
            self._ex_events = [e for e in self._ex_events if e is not ev][-15:] + [ev]
            t4 = time.perf_counter() if tr else 0.0
            # (not torch's copy_: its pinned-memory bookkeeping now and then took 7 ms on the host inside a pipelined
            # loop — three of ten 20-step runs lost a third of their time to one such call)
            if not self._zero_copy:
                B._check(B.lib().tq_copy_to_host_async(
                    self.dev.ctx, int(self.device), self._host_all.data_ptr(), self._merged_all.data_ptr(),
                    self._merged_all.numel() * 4, C.c_void_p(self.stream)))
            t5 = time.perf_counter() if tr else 0.0
        if tr and time.perf_counter() - t0 > 0.003:
            t6 = time.perf_counter()
            print("[runner] enter-stream %.2f collect %.2f ev0 %.2f exchange+merge %.2f ev1 %.2f copy %.2f exit-stream %.2f ms" %
                  tuple((b - a) * 1e3 for a, b in ((t0, t0b), (t0b, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))),
                  file=sys.stderr)

    def synchronize(self):
        self.stream_obj.synchronize()

    def results(self):
        """(scores, segment_ords, docs, counts) of the last finished step (host copies)."""
        return [h.clone().numpy() for h in self.host]

    def batch_stats(self):
        """Per-step statistics summed over the local segments (kernel_ms: the segments' scan
        kernels run back to back on one stream)."""
        out = None
        for s in range(self.n_local):
            st = self.dev.last_batch_stats(s)
            if out is None:
                out = dict(st)
            else:
                for key in ("algorithmic_bytes", "matches", "kernel_ms", "total_ms", "tiles", "chunks",
                            "host_plan_ms", "unique_bytes"):
                    out[key] += st[key]
                out["kernel_mask"] |= st["kernel_mask"]
                out["kernels"] = sorted(set(out["kernels"]) | set(st["kernels"]))
        return out

    def exchange_ms(self):
        """Mean GPU time of [all-gather + merge_top_k] over the last (<= 16) finished steps."""
        if not self._ex_events:
            return 0.0
        self.stream_obj.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._ex_events]
        self._ex_events = []
        return float(sum(ms) / len(ms))

    def close(self):
        self.dev.close()
