// tq_submit.cpp — tq_submit / tq_wait / tq_search_one: single queries of concurrent callers coalesced into
// batched launches (leader / followers)
// Part of the C ABI library of include/tantivy_amd.h (internal declarations: tq_internal.hpp).
#include "tq_internal.hpp"

// ---- concurrent single-query entry: tq_submit / tq_wait / tq_search_one
// tantivy lets any number of threads call Searcher::search at once, one query per call
// (src/core/searcher.rs:180-238; Weight is Send + Sync, src/query/weight.rs:66), and each call ends
// in one collect_segment per segment (src/collector/mod.rs:173-183).  One query per launch is the
// 0.3 ms / 3 k queries/s regime of this device; the batched launch needs the queries of MANY callers.
// Leader / followers: a caller puts its query on the segment's pending list; whoever waits while no
// batch is running becomes the leader, takes everything pending (callers keep arriving while the
// previous batch runs: that IS the batching), runs it as one tq_search_batch under the segment
// lock, hands every caller its rows and wakes them up.  Nobody waits for a batch to fill.
struct tq_ticket {
  tq_segment *seg = nullptr;
  tq_query q{};
  CallOpts co{};
  float *out_scores = nullptr;
  uint32_t *out_docs = nullptr, *out_count = nullptr;
  int rc = TQ_OK;
  std::string err;
  // its caller sleeps on its OWN condition variable: a finished batch wakes its callers and ONE thread to lead the
  // next batch — with one shared variable and notify_all every completion woke every waiter of the segment (1 024
  // threads on 16 cores: 35 k q/s at p99 94 ms, most of it the herd taking the queue's mutex in turn)
  // ... under its OWN mutex (round 6): a caller whose rows are in goes home without touching the queue's mutex — with
  // the queue's mutex behind every ticket's variable the few hundred callers of a finished batch still queued up for
  // it one by one, in front of the callers trying to submit the next batch's queries.
  std::mutex m;
  std::condition_variable cv;
  bool signal = false;     // (m) look again: the rows are in (done), or somebody has to lead
  bool done = false;       // (m) verdict and rows delivered
  bool finishing = false;  // (Q.m) its batch is over, delivery is on its way: nothing left to lead or to register for
  bool waiting = false;    // (Q.m) its caller is blocked in tq_wait: the ticket is linked into Q's waiting list
  tq_ticket *wprev = nullptr, *wnext = nullptr;
};
struct SubmitQueue {
  std::mutex m;
  std::condition_variable cv_arrive;  // a query was submitted (the leader's arrival window)
  std::deque<tq_ticket *> pending;
  tq_ticket *whead = nullptr, *wtail = nullptr;  // tickets whose callers are blocked in tq_wait, oldest first
  void wait_link(tq_ticket *t) {
    if (t->waiting) return;
    t->waiting = true;
    t->wprev = wtail;
    t->wnext = nullptr;
    (wtail ? wtail->wnext : whead) = t;
    wtail = t;
  }
  void wait_unlink(tq_ticket *t) {
    if (!t->waiting) return;
    t->waiting = false;
    (t->wprev ? t->wprev->wnext : whead) = t->wnext;
    (t->wnext ? t->wnext->wprev : wtail) = t->wprev;
    t->wprev = t->wnext = nullptr;
  }
  bool leader_active = false;
  size_t last_batch = 0;  // queries the previous batch carried
  tq_submit_stats stats{};
  // the leader's scratch (the synchronous road: a batch that has to be bisected)
  std::vector<tq_query> qs;
  std::vector<float> sc;
  std::vector<uint32_t> dc, ct;
  // Two coalesced batches can be in flight: the leader of batch i hands leadership on as soon as the batch is
  // ENQUEUED, waits for its rows off the lock, and the next leader plans and enqueues batch i + 1 behind it on the
  // segment's stream meanwhile (the library pipelines two batches: two staging buffers).
  static constexpr int kSlots = 2;
  tqi::HostBatchSlot slot[kSlots];
  std::vector<tq_query> slot_qs[kSlots];
  bool slot_busy[kSlots] = {false, false};
  std::condition_variable cv_slot;
};
void tq_free_submit_queue(SubmitQueue *q) {
  if (!q) return;
  for (tqi::HostBatchSlot &sl : q->slot) {  // (the segment is idle: tq_segment_free has drained its streams)
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.done_blocking) (void)hipEventDestroy(sl.done_blocking);
    sl.done = sl.done_blocking = nullptr;
    sl.out.release();
  }
  delete q;
}
SubmitQueue *tq_new_submit_queue() { return new (std::nothrow) SubmitQueue(); }

namespace tqi {
constexpr size_t kSubmitMaxBatch = 16384;

// The queue exists from segment upload on (tq_api.cpp: tq_new_submit_queue): no process-wide mutex on the
// submit path, no lazy creation to race on.
SubmitQueue *submit_queue(tq_segment *s) { return s->submit; }

// one launch for tickets [lo, hi) of `batch` (same options); rows go to the callers' buffers.
// A batch-level failure is split only when it can be one query's fault — an unsupported shape, a bad
// argument, a corrupt list (TQ_ERR_UNSUPPORTED / INVALID / FORMAT): the batch is bisected, so one bad query
// among n costs about 2 log2(n) extra launches and every caller gets its own verdict.  A device error
// (TQ_ERR_HIP, out of memory) fails the whole batch once: re-running thousands of queries one at a time
// against a device that cannot allocate would stall every caller of the segment for seconds.
void run_ticket_range(SubmitQueue &Q, tq_segment *s, std::vector<tq_ticket *> &batch, size_t lo, size_t hi) {
  const uint32_t n = (uint32_t)(hi - lo);
  if (!n) return;
  uint32_t stride = 1;
  Q.qs.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    Q.qs[i] = batch[lo + i]->q;
    stride = std::max(stride, batch[lo + i]->q.k);
  }
  Q.sc.resize((size_t)n * stride);
  Q.dc.resize((size_t)n * stride);
  Q.ct.resize(n);
  int rc;
  {
    TQ_SEGMENT_LOCK(s);
    rc = search_batch_host(s, Q.qs.data(), n, stride, Q.sc.data(), Q.dc.data(), Q.ct.data(), batch[lo]->co);
  }
  if (rc == TQ_OK) {
    for (uint32_t i = 0; i < n; ++i) {
      tq_ticket *t = batch[lo + i];
      const uint32_t k = t->q.k;
      memcpy(t->out_scores, Q.sc.data() + (size_t)i * stride, k * sizeof(float));
      memcpy(t->out_docs, Q.dc.data() + (size_t)i * stride, k * sizeof(uint32_t));
      *t->out_count = Q.ct[i];
      t->rc = TQ_OK;
    }
    return;
  }
  const bool per_query = rc == TQ_ERR_UNSUPPORTED || rc == TQ_ERR_INVALID || rc == TQ_ERR_FORMAT;
  if (n == 1 || !per_query) {
    const std::string err = g_last_error;
    for (size_t i = lo; i < hi; ++i) {
      batch[i]->rc = rc;
      batch[i]->err = err;
    }
    return;
  }
  const size_t mid = lo + n / 2;
  run_ticket_range(Q, s, batch, lo, mid);
  run_ticket_range(Q, s, batch, mid, hi);
}

// Never throws: the planner allocates (std::vector, PodVec::reserve) and an exception escaping the leader
// would leave leader_active set for good — every waiter of the segment deadlocked — and cross the extern "C"
// boundary.  Whatever was not answered is failed.
void run_ticket_batch(SubmitQueue &Q, tq_segment *s, std::vector<tq_ticket *> &batch) noexcept {
  for (tq_ticket *t : batch) t->rc = TQ_ERR_HIP;  // (overwritten by every verdict)
  try {
    run_ticket_range(Q, s, batch, 0, batch.size());
  } catch (const std::exception &e) {
    for (tq_ticket *t : batch)
      if (t->rc != TQ_OK) {
        t->rc = TQ_ERR_HIP;
        t->err = std::string("tq_wait: ") + e.what();
      }
  } catch (...) {
    for (tq_ticket *t : batch)
      if (t->rc != TQ_OK) {
        t->rc = TQ_ERR_HIP;
        t->err = "tq_wait: unknown exception in the batch";
      }
  }
}

// parked: the ticket was linked into the waiting list when it was submitted (tq_search_one, somebody was leading): its
// caller goes to sleep without taking the queue's mutex a second time
int ticket_wait(tq_ticket *t, bool parked = false) {
  tq_segment *s = t->seg;
  SubmitQueue &Q = *s->submit;
  std::unique_lock<std::mutex> lk(Q.m, std::defer_lock);
  if (!parked) lk.lock();
  std::vector<tq_ticket *> batch;
  auto leave_waiting = [&]() { Q.wait_unlink(t); };
  // (under t->m) verdict and rows to a ticket's caller
  auto deliver = [](tq_ticket *b) {
    std::lock_guard<std::mutex> bl(b->m);
    b->done = true;
    b->signal = true;
    b->cv.notify_one();
  };
  bool own_done = false;  // this caller's ticket was answered by the batch it led itself
  while (!own_done) {
    if (parked || t->finishing || Q.leader_active || Q.pending.empty()) {
      if (!parked) {
        if (!t->finishing) Q.wait_link(t);
        lk.unlock();
      }
      parked = false;
      bool finished;
      {
        std::unique_lock<std::mutex> tl(t->m);
        t->cv.wait(tl, [&] { return t->signal; });
        t->signal = false;
        finished = t->done;
      }
      if (finished) {  // (whoever finished the batch unlinked the ticket before delivering: nothing of the queue is touched)
        const int rc = t->rc;
        if (rc != TQ_OK) g_last_error = t->err;
        return rc;
      }
      lk.lock();
      continue;
    }
    leave_waiting();
    // lead one batch: everything pending that runs under the first ticket's options.  Callers of
    // the batch that just finished are on their way back with their next query: the leader gives them
    // up to submit_window_us to arrive (until as many are pending as the last batch carried) — without
    // it the first caller back leads a batch of one and everybody else waits a whole launch longer
    Q.leader_active = true;
    const int window_us = __atomic_load_n(&s->opt.submit_window_us, __ATOMIC_RELAXED);
    if (Q.pending.size() < Q.last_batch && window_us > 0) {
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us);
      while (Q.pending.size() < Q.last_batch && Q.pending.size() < kSubmitMaxBatch)
        if (Q.cv_arrive.wait_until(lk, deadline) == std::cv_status::timeout) break;
    }
    batch.clear();
    const CallOpts co = Q.pending.front()->co;
    for (auto it = Q.pending.begin(); it != Q.pending.end() && batch.size() < kSubmitMaxBatch;) {
      if ((*it)->co.exhaustive == co.exhaustive && (*it)->co.bound_slack == co.bound_slack) {
        batch.push_back(*it);
        it = Q.pending.erase(it);
      } else {
        ++it;
      }
    }
    Q.last_batch = batch.size();
    // wake-ups (under the lock): two blocked callers whose queries are still open, the first to take the mutex leads
    // the next batch (callers that are not blocked yet lead themselves when they arrive and find nobody leading).
    // 1 024 threads: 35 k q/s at p99 94 ms with notify_all on one shared variable, 56-85 k at p99 25-80 ms with
    // 1 / 2 / 4 / 64 woken; 16 and 64 threads do not care
    auto wake_a_leader = [&]() {
      static const uint32_t kWake = std::max<uint32_t>(1u, tune_u32("TQ_SUBMIT_WAKE", 2));
      if (Q.pending.empty()) return;
      uint32_t woken = 0;
      for (tq_ticket *w = Q.whead; w; w = w->wnext)
        if (w != t) {  // (a linked ticket is open: finish() unlinks what it answers)
          {
            std::lock_guard<std::mutex> wl(w->m);
            w->signal = true;
            w->cv.notify_one();
          }
          if (++woken == kWake) break;
        }
    };
    // the batch's verdicts are in.  Under the lock: its tickets leave the queue's lists; off the lock (finish_deliver):
    // their callers are woken — a few hundred wake-ups are a few hundred microseconds during which the next batch's
    // queries must be able to arrive
    auto finish = [&]() {
      ++Q.stats.batches;
      Q.stats.queries += batch.size();
      Q.stats.max_batch = std::max<uint64_t>(Q.stats.max_batch, batch.size());
      for (tq_ticket *b : batch) {
        Q.wait_unlink(b);
        b->finishing = true;
      }
    };
    auto finish_deliver = [&]() {
      for (tq_ticket *b : batch) {
        if (b == t)
          own_done = true;
        else
          deliver(b);  // (b may be gone the moment this returns)
      }
    };
    static const bool kOverlap = tune_u32("TQ_SUBMIT_OVERLAP", 1) != 0;
    int si = -1;
    if (kOverlap) {
      while (Q.slot_busy[0] && Q.slot_busy[1]) Q.cv_slot.wait(lk);
      si = Q.slot_busy[0] ? 1 : 0;
      Q.slot_busy[si] = true;
    }
    lk.unlock();
    bool enqueued = false;
    if (si >= 0) {  // plan + enqueue under the segment lock, then let the next leader in
      int brc = TQ_ERR_HIP;
      for (tq_ticket *b : batch) b->rc = TQ_ERR_HIP;
      try {
        std::vector<tq_query> &qs = Q.slot_qs[si];
        qs.resize(batch.size());
        uint32_t stride = 1;
        for (size_t i = 0; i < batch.size(); ++i) {
          qs[i] = batch[i]->q;
          stride = std::max(stride, batch[i]->q.k);
        }
        TQ_SEGMENT_LOCK(s);
        brc = search_batch_host_begin(s, qs.data(), (uint32_t)qs.size(), stride, co, Q.slot[si]);
      } catch (...) {
        brc = TQ_ERR_HIP;
      }
      enqueued = brc == TQ_OK;
    }
    if (enqueued) {
      // The batch is on the stream.  If a batch's worth of queries is already waiting (hundreds of callers), the next
      // leader may form and enqueue it behind this one right away; with few callers the ones that matter are IN this
      // batch, and a leader let in now would launch the handful left over (16 threads: 6 queries per launch instead
      // of 14, 21 k instead of 28 k q/s) — leadership is then handed on when the rows are back, as before.
      // (256 threads: 113 k q/s one batch at a time, 139 k handing over always, 122 / 132 / 130 k with a threshold of
      // a half / a quarter / a sixteenth of the batch, at least 32; 64 threads 68 / 58 / 72 / 73 / 67 k)
      static const size_t kOverlapMin = std::max<uint32_t>(1u, tune_u32("TQ_SUBMIT_OVERLAP_MIN", 32));
      lk.lock();
      static const size_t kOverlapDiv = std::max<uint32_t>(1u, tune_u32("TQ_SUBMIT_OVERLAP_DIV", 4));
      const bool handed_over = Q.pending.size() >= std::max(kOverlapMin, batch.size() / kOverlapDiv);
      if (handed_over) {
        Q.leader_active = false;
        wake_a_leader();
      }
      lk.unlock();
      tqi::HostBatchSlot &S = Q.slot[si];
      // (nothing may leave this block by exception: the slot, the tickets and — without a hand-over — the leadership
      // are released below; a bad_alloc here used to unwind through tq_wait with all three still held, and every
      // caller of the batch waited for good.  ADVICE r05)
      int erc = TQ_ERR_HIP;
      try {
        erc = search_batch_host_end(s, S);
      } catch (...) {
        erc = TQ_ERR_HIP;
      }
      for (size_t i = 0; i < batch.size(); ++i) {
        tq_ticket *b = batch[i];
        if (erc == TQ_OK) {
          const uint8_t *h = (const uint8_t *)S.out.p;
          const uint32_t k = b->q.k;
          memcpy(b->out_scores, h + ((size_t)i * S.stride) * sizeof(float), k * sizeof(float));
          memcpy(b->out_docs, h + S.o_docs + ((size_t)i * S.stride) * sizeof(uint32_t), k * sizeof(uint32_t));
          *b->out_count = ((const uint32_t *)(h + S.o_counts))[i];
          b->rc = TQ_OK;
        } else {
          b->rc = erc;
          try {
            b->err = g_last_error;
          } catch (...) {  // (the status code still reaches the caller)
          }
        }
      }
      lk.lock();
      Q.slot_busy[si] = false;
      Q.cv_slot.notify_one();
      finish();
      if (!handed_over) Q.leader_active = false;
      if (!Q.leader_active) wake_a_leader();  // (whoever was woken at hand-over may have been in this batch)
      lk.unlock();
      finish_deliver();
      lk.lock();
      continue;
    }
    // the synchronous road: no slot (TQ_SUBMIT_OVERLAP=0), or the batch could not be enqueued — a query the device
    // refuses, a bad argument: the batch is bisected, every caller gets its own verdict (this leader stays the leader)
    run_ticket_batch(Q, s, batch);
    lk.lock();
    if (si >= 0) {
      Q.slot_busy[si] = false;
      Q.cv_slot.notify_one();
    }
    finish();
    Q.leader_active = false;
    wake_a_leader();
    lk.unlock();
    finish_deliver();
    lk.lock();
  }
  leave_waiting();
  const int rc = t->rc;
  if (rc != TQ_OK) g_last_error = t->err;  // (this thread's slot)
  return rc;
}
}  // namespace tqi

extern "C" {

}  // extern "C"
namespace tqi {
// tq_submit; park (tq_search_one: the caller waits right away): if somebody is leading, the ticket joins the waiting
// list under the same lock that puts it on the pending list — one acquisition of the queue's mutex per query, not two
int submit_ticket(tq_segment *s, const tq_query *q, const tq_search_opts *opts, float *out_scores,
                  uint32_t *out_docs, uint32_t *out_count, tq_ticket **out, bool *park) {
  if (!s || !q || !out_scores || !out_docs || !out_count || !out)
    return fail(TQ_ERR_INVALID, "tq_submit: null argument");
  // what can be judged without the segment's state is judged here: a bad query never joins a batch
  if (q->n_terms == 0 || q->n_terms > TQ_MAX_TERMS)
    return fail(TQ_ERR_INVALID, "tq_submit: n_terms %u not in 1..%u", q->n_terms, TQ_MAX_TERMS);
  if (q->k == 0 || q->k > TQ_MAX_K) return fail(TQ_ERR_INVALID, "tq_submit: k %u not in 1..%u", q->k, TQ_MAX_K);
  if (!q->terms || !q->weights || !q->tf_cache) return fail(TQ_ERR_INVALID, "tq_submit: null terms/weights/tf_cache");
  if (!s->submit) return fail(TQ_ERR_INVALID, "tq_submit: segment without a submit queue");
  // (no segment lock here: the leader of the running batch holds it for the whole launch, and a submit
  // must not wait for that — resolve_opts reads the two option words with atomic loads)
  CallOpts co;
  {
    const int rc = resolve_opts(s, opts, co);
    if (rc != TQ_OK) return rc;
  }
  tq_ticket *t = new (std::nothrow) tq_ticket();
  if (!t) return fail(TQ_ERR_INVALID, "tq_submit: out of memory");
  t->seg = s;
  t->q = *q;
  t->co = co;
  t->out_scores = out_scores;
  t->out_docs = out_docs;
  t->out_count = out_count;
  SubmitQueue *Q = submit_queue(s);
  {
    std::lock_guard<std::mutex> lk(Q->m);
    Q->pending.push_back(t);
    if (park && Q->leader_active) {
      Q->wait_link(t);
      *park = true;
    }
  }
  Q->cv_arrive.notify_one();  // (a leader may be holding its batch open for this query)
  *out = t;
  return TQ_OK;
}
}  // namespace tqi
extern "C" {

int tq_submit(tq_segment *s, const tq_query *q, const tq_search_opts *opts, float *out_scores,
              uint32_t *out_docs, uint32_t *out_count, tq_ticket **out) {
  return tqi::submit_ticket(s, q, opts, out_scores, out_docs, out_count, out, nullptr);
}

int tq_wait(tq_ticket *t) {
  if (!t) return fail(TQ_ERR_INVALID, "tq_wait: null ticket");
  const int rc = ticket_wait(t);
  delete t;
  return rc;
}

int tq_search_one(tq_segment *s, const tq_query *q, const tq_search_opts *opts, float *out_scores,
                  uint32_t *out_docs, uint32_t *out_count) {
  tq_ticket *t = nullptr;
  bool parked = false;
  const int rc = tqi::submit_ticket(s, q, opts, out_scores, out_docs, out_count, &t, &parked);
  if (rc != TQ_OK) return rc;
  const int wrc = tqi::ticket_wait(t, parked);
  delete t;
  return wrc;
}

int tq_get_submit_stats(tq_segment *s, tq_submit_stats *out, int reset) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_get_submit_stats: null argument");
  SubmitQueue *Q = submit_queue(s);
  if (!Q) return fail(TQ_ERR_INVALID, "tq_get_submit_stats: segment without a submit queue");
  std::lock_guard<std::mutex> lk(Q->m);
  *out = Q->stats;
  if (reset) Q->stats = tq_submit_stats{};
  return TQ_OK;
}

}  // extern "C"

