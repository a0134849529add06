"""The submit queue (tantivy_amd/csrc/tq_submit.cpp: tq_submit / tq_wait / tq_search_one) without a GPU: the real
translation unit built with -fsanitize=thread around a stand-in for search_batch_host (rows derived from each query, a
short sleep for the launch, TQ_ERR_UNSUPPORTED for a batch that holds a marked query), 64 threads x 250 single-query
calls.  Checks: every caller gets ITS rows, a refused query fails alone (the batch is bisected), calls are coalesced,
nothing deadlocks, ThreadSanitizer reports no race — the per-ticket wake-ups of round 5 (a finished batch wakes its
callers and two candidates to lead the next one) and the hand-over of leadership once a batch is enqueued (two coalesced
batches in flight: search_batch_host_begin / _end)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

SRC = r'''
#include "tq_internal.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
namespace tqi {
static std::atomic<uint64_t> g_launches{0}, g_queries{0};
// stand-in for the launch: rows derived from the query itself, one marked query refuses its whole batch
int fake_search_batch_host(tq_segment *, const tq_query *queries, uint32_t n, uint32_t stride, float *sc, uint32_t *dc,
                           uint32_t *ct, const CallOpts &) {
  for (uint32_t i = 0; i < n; ++i)
    if (queries[i].terms[0] % 97u == 0u) return fail(TQ_ERR_UNSUPPORTED, "query %u refused", i);
  std::this_thread::sleep_for(std::chrono::microseconds(60));
  for (uint32_t i = 0; i < n; ++i) {
    for (uint32_t j = 0; j < queries[i].k; ++j) {
      sc[(size_t)i * stride + j] = (float)(queries[i].terms[0] % 1000u) + (float)j;
      dc[(size_t)i * stride + j] = queries[i].terms[0] * 16u + j;
    }
    ct[i] = queries[i].k;
  }
  ++g_launches;
  g_queries += n;
  return TQ_OK;
}
// the two halves of the pipelined road: _begin "enqueues" (rows computed into the slot's buffer, plain memory here),
// _end is the wait for the device
int fake_begin(tq_segment *s, const tq_query *queries, uint32_t n, uint32_t stride, const CallOpts &co, HostBatchSlot &slot) {
  const size_t cells = (size_t)n * stride;
  slot.o_docs = (cells * sizeof(float) + 255) & ~(size_t)255;
  slot.o_counts = (slot.o_docs + cells * sizeof(uint32_t) + 255) & ~(size_t)255;
  slot.n = n;
  slot.stride = stride;
  const size_t need = slot.o_counts + (size_t)n * sizeof(uint32_t);
  if (slot.out.cap < need) {
    slot.out.p = realloc(slot.out.p, need);
    slot.out.cap = need;
  }
  uint8_t *h = (uint8_t *)slot.out.p;
  for (uint32_t i = 0; i < n; ++i)
    if (queries[i].terms[0] % 97u == 0u) return fail(TQ_ERR_UNSUPPORTED, "query %u refused", i);
  for (uint32_t i = 0; i < n; ++i) {
    for (uint32_t j = 0; j < queries[i].k; ++j) {
      ((float *)h)[(size_t)i * stride + j] = (float)(queries[i].terms[0] % 1000u) + (float)j;
      ((uint32_t *)(h + slot.o_docs))[(size_t)i * stride + j] = queries[i].terms[0] * 16u + j;
    }
    ((uint32_t *)(h + slot.o_counts))[i] = queries[i].k;
  }
  (void)s;
  (void)co;
  ++g_launches;
  g_queries += n;
  return TQ_OK;
}
int fake_end(tq_segment *, HostBatchSlot &) {
  std::this_thread::sleep_for(std::chrono::microseconds(60));
  return TQ_OK;
}
}  // namespace tqi
#define search_batch_host fake_search_batch_host
#define search_batch_host_begin fake_begin
#define search_batch_host_end fake_end
#include "tq_submit.cpp"
#undef search_batch_host
#undef search_batch_host_begin
#undef search_batch_host_end

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 250;
  tq_segment *s = new tq_segment();
  s->submit = tq_new_submit_queue();
  s->opt.submit_window_us = argc > 3 ? atoi(argv[3]) : 100;
  static float cache[256];
  std::atomic<int> bad{0}, refused{0};
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      for (int i = 0; i < N; ++i) {
        tq_term_handle term = (tq_term_handle)(t * 100000 + i + 1);
        float w = 1.0f;
        tq_query q{};
        q.n_terms = 1;
        q.terms = &term;
        q.weights = &w;
        q.tf_cache = cache;
        q.mode = TQ_MODE_OR;
        q.k = 1 + (uint32_t)((t + i) % 7);
        float sc[8];
        uint32_t dc[8], n = 0;
        int rc;
        if (i % 3 == 0) {  // submit first, wait later (two tickets in flight)
          tq_ticket *a = nullptr, *b = nullptr;
          tq_term_handle term2 = term + 50000;
          tq_query q2 = q;
          q2.terms = &term2;
          float sc2[8];
          uint32_t dc2[8], n2 = 0;
          rc = tq_submit(s, &q, nullptr, sc, dc, &n, &a);
          int rc2 = tq_submit(s, &q2, nullptr, sc2, dc2, &n2, &b);
          if (rc == TQ_OK) rc = tq_wait(a);
          if (rc2 == TQ_OK) rc2 = tq_wait(b);
          const bool want_fail2 = term2 % 97u == 0u;
          if ((rc2 != TQ_OK) != want_fail2) ++bad;
          if (rc2 == TQ_OK && (n2 != q2.k || dc2[0] != term2 * 16u)) ++bad;
        } else {
          rc = tq_search_one(s, &q, nullptr, sc, dc, &n);
        }
        const bool want_fail = term % 97u == 0u;
        if ((rc != TQ_OK) != want_fail) ++bad;
        if (rc != TQ_OK) { ++refused; continue; }
        if (n != q.k) ++bad;
        for (uint32_t j = 0; j < n; ++j)
          if (dc[j] != term * 16u + j || sc[j] != (float)(term % 1000u) + (float)j) ++bad;
      }
    });
  for (auto &x : th) x.join();
  tq_submit_stats st{};
  tq_get_submit_stats(s, &st, 0);
  std::printf("threads %d x %d: %llu launches, %llu queries ok, %d refused, queue batches %llu (max %llu), bad %d\n", T, N,
              (unsigned long long)tqi::g_launches.load(), (unsigned long long)tqi::g_queries.load(), refused.load(),
              (unsigned long long)st.batches, (unsigned long long)st.max_batch, bad.load());
  return bad.load() ? 1 : 0;
}
'''


@pytest.mark.parametrize("sanitize", ["thread", "none"])
def test_submit_queue_stress(tmp_path_factory, sanitize):
    from tantivy_amd import build as B

    B.build()
    d = tmp_path_factory.mktemp("submitq_" + sanitize)
    src = d / "submit_stress.cpp"
    src.write_text(SRC)
    csrc = os.path.join(ROOT, "tantivy_amd", "csrc")
    flags = ["-fsanitize=thread"] if sanitize == "thread" else []
    obj = str(d / "submit_stress.o")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function",
                        "-I", csrc, "-I", os.path.join(ROOT, "include")] + flags + ["-c", str(src), "-o", obj],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and sanitize == "thread":
        pytest.skip("no ThreadSanitizer build here: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s) + ".o") for s in B.SOURCES
              if os.sep + "csrc" + os.sep in s and os.path.basename(s) != "tq_submit.cpp"]
    exe = str(d / "submit_stress")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950"] + flags + ["-o", exe, obj] + others + ["-ldl", "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and sanitize == "thread":
        pytest.skip("ThreadSanitizer runtime not linkable here: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    for args, env_extra in ((["64", "250", "100"], {}), (["16", "400", "0"], {}), (["200", "60", "30"], {}),
                            (["64", "120", "100"], {"TQ_SUBMIT_OVERLAP": "0"})):  # (0: one batch at a time, the synchronous road)
        r = subprocess.run([exe] + args, env=dict(os.environ, **env_extra), capture_output=True, text=True,
                           timeout=600)  # (a deadlock is a timeout)
        out = r.stdout + r.stderr
        assert r.returncode == 0, out[-3000:]
        assert "ThreadSanitizer" not in out, out[-3000:]
        assert "bad 0" in out, out
        # coalescing happened: fewer launches than queries answered (window 0 still batches what queued up meanwhile)
        launches = int(out.split(" launches")[0].split(": ")[-1])
        queries = int(out.split(" queries ok")[0].split(", ")[-1])
        assert queries > 0 and launches < queries, out
