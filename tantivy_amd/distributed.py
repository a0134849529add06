"""Segment-per-rank execution: every rank holds one segment, all ranks run the
same query batch with the same global BM25 statistics, and the per-segment top-k lists are
exchanged with ONE all-gather (RCCL over xGMI on GPUs, gloo on CPU) before `merge_top_k`
(src/collector/sort_key_top_collector.rs:76-95) picks the global top-(offset+limit) by
(score desc, segment_ord asc, doc asc).  No other collective exists on this path: BM25
statistics are sums of per-segment counters known to the host before dispatch (bm25.rs:27-50).
"""
import ctypes as C

import numpy as np

from . import binding as B


def allgather_topk(scores, docs, counts, group=None):
    """scores/docs: [n_queries, k] tensors, counts: [n_queries] (any device).  Returns the
    gathered [world, n_queries, k] / [world, n_queries] tensors (rank r == segment_ord r)."""
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


def merge_gathered_device(ctx, device, g_scores, g_docs, g_counts, offset, limit, stream=None):
    """merge_top_k on the device (tq_merge_topk_device); tensors stay on the GPU."""
    import torch

    S, n, k = g_scores.shape
    dev = g_scores.device
    out_s = torch.empty((n, limit), dtype=torch.float32, device=dev)
    out_o = torch.empty((n, limit), dtype=torch.int32, device=dev)
    out_d = torch.empty((n, limit), dtype=torch.int32, device=dev)
    out_c = torch.empty(n, dtype=torch.int32, device=dev)
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    B._check(B.lib().tq_merge_topk_device(
        ctx, int(device), g_scores.data_ptr(), g_docs.data_ptr(), g_counts.data_ptr(), None, S, n, k,
        int(offset), int(limit), out_s.data_ptr(), out_o.data_ptr(), out_d.data_ptr(),
        out_c.data_ptr(), C.c_void_p(st)))
    return out_s, out_o, out_d, out_c
