#!/usr/bin/env python
"""T host threads x single-query Searcher::search calls (tq_search_one coalescing) on the headline
stream: throughput, p50 / p99 latency and queries per launch by arrival window (submit_window_us)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import tantivy_amd  # noqa: E402

import resource  # noqa: E402


def cpu_now():
    r = resource.getrusage(resource.RUSAGE_SELF)
    global last_sys
    last_sys = r.ru_stime
    return r.ru_utime + r.ru_stime


def throttled_now():
    """microseconds this cgroup has been throttled so far (CFS bandwidth control), 0 if unknown"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if k == "throttled_usec":
                    return int(v)
                if k == "throttled_time":
                    return int(v) // 1000
        except OSError:
            pass
    return 0


seg = O.synth_segment(10_000_000, n_terms=256)
dev = tantivy_amd.DeviceIndex([seg], devices=[0])
ids = O.zipf_queries(10000, 2, 256, seed=20260921)
qs = [(O.MODE_AND, q.tolist()) for q in ids]
dev.set_option("exhaustive", 0)
for window in [int(x) for x in os.environ.get("WINDOWS", "100,0,30,300").split(",")]:
    dev.set_option("submit_window_us", window)
    for nthreads in [int(x) for x in os.environ.get("THREADS", "16,64,256").split(",")]:
        n = nthreads * int(os.environ.get("PER_THREAD", "100"))
        run = (qs * (n // len(qs) + 1))[:n]  # (the stream repeated: every thread gets PER_THREAD queries)
        dev.search_concurrent(run[:4 * nthreads], 10, nthreads)
        dev.submit_stats(reset=True)
        cpu0, thr0 = cpu_now(), throttled_now()
        sys0 = last_sys
        _, _, _, _, lat_ms, wall_ms = dev.search_concurrent(run, 10, nthreads)
        cpu_us = (cpu_now() - cpu0) * 1e6 / n
        sys_us = (last_sys - sys0) * 1e6 / n
        thr_ms = (throttled_now() - thr0) / 1e3
        st = dev.submit_stats()
        v = sorted(float(x) for x in lat_ms)
        print("window %4d us threads %4d: %8.0f q/s p50 %.3f ms p99 %.3f ms, %.1f queries per launch (max %d), wall %.0f ms, "
              "%.1f us of CPU per query (%.1f in the kernel; %.1f cores busy), cgroup throttled %.0f ms" %
              (window, nthreads, n / (wall_ms * 1e-3), v[len(v) // 2], v[int(len(v) * 0.99)],
               st["queries"] / max(1, st["batches"]), st["max_batch"], wall_ms, cpu_us, sys_us, cpu_us * n / (wall_ms * 1e3), thr_ms))
dev.close()
