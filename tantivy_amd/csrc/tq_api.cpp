// tq_api.cpp — the C ABI of include/tantivy_amd.h: segment residency, skip-list unrolling,
// batch planning (tiles / chunks / partial lists) and kernel launches.  Compiled with hipcc.
//
// Host-side format walkers restate (file:line under the tantivy checkout):
//   skip entries        src/postings/skip.rs:205-253,275-302
//   list framing        src/postings/block_segment_postings.rs:78-88,107-116
//   vint tail           src/postings/compression/vint.rs:44-108
//   positions framing   src/positions/reader.rs:43-56,84-101
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <thread>
#include <vector>

#include "../../include/tantivy_amd.h"
#include "tq_device.h"
#include "tq_launch.h"
#include "tq_prepare.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess)                                                               \
      return fail(TQ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),    \
                  __FILE__, __LINE__);                                                   \
  } while (0)

constexpr size_t PAD = 1088;  // over-read slack after every device byte buffer (staged block loads)

// A grow-only device buffer.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return TQ_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    size_t ncap = std::max(n, cap * 2);
    HIP_TRY(hipMalloc(&p, ncap));
    cap = ncap;
    return TQ_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};
struct PinnedBuf {
  void *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return TQ_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    size_t ncap = std::max(n, cap * 2);
    HIP_TRY(hipHostMalloc(&p, ncap, hipHostMallocDefault));
    cap = ncap;
    return TQ_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct TermHost {
  void *blob = nullptr;  // one device allocation holding every per-term array
  void *dense_blob = nullptr;  // bitmap + rank directory of a dense list
  void *posdir_blob = nullptr; // position directory of a dense list with positions
  void *tf8_blob = nullptr;    // term freqs of a dense list as bytes (posting index -> min(tf, 255))
  void *pos_blob = nullptr;    // device-side prepare: positions tables (sized after the walk)
  void *flat_blob = nullptr;   // a list without a bitmap as plain arrays (doc ids | byte-wide tfs), built on
                               // first use by an unpruned union batch (tq_xunion.hip)
  uint32_t doc_freq = 0, n_blocks = 0, n_full = 0, n_tail = 0;
  uint32_t last_doc = 0;
  bool wants_col = true;  // false: the segment's columns are reserved for other lists
  uint64_t postings_len = 0, positions_len = 0;
  uint64_t n_positions = 0;
};

inline uint32_t rd32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
// common VInt (common/src/vint.rs:61-112): stop bit on the LAST byte
bool read_vint(const uint8_t *d, size_t len, size_t &at, uint64_t &out) {
  uint64_t r = 0;
  unsigned shift = 0;
  while (at < len) {
    uint8_t b = d[at++];
    r |= (uint64_t)(b & 127u) << shift;
    if (b & 128u) {
      out = r;
      return true;
    }
    shift += 7;
    if (shift > 63) return false;
  }
  return false;
}
bool read_vint32_block(const uint8_t *d, size_t len, size_t &at, uint32_t &out) {
  uint32_t r = 0, shift = 0;
  while (at < len) {
    uint8_t b = d[at++];
    r += (uint32_t)(b & 127u) << shift;
    if (b & 128u) {
      out = r;
      return true;
    }
    shift += 7;
  }
  return false;
}

struct Options {
  int exhaustive = 0;  // 0 = block-max pruned top-k (what the reference executes), 1 = score every match
  int timing = 0;
  int use_dpp = 1;
  int dense = 1;      // build bitmaps for dense lists at tq_term_prepare
  int dense_ratio = TQD_DENSE_RATIO;  // ... for lists with doc_freq >= max_doc / dense_ratio
  int dense_budget_x = 8;  // ... while bitmaps + byte-wide tfs + doc matrix + signatures + position directories
                           // stay below this multiple of the segment's bytes
  int use_dense = 1;  // let the scan kernels use them
  int docmat = 1;     // also build the doc-major matrix of the dense lists
  int docsig = 1;     // ... and the per-doc signature word of the lists without a column
  int device_prepare = 0;  // walk skip lists / build dense tables on the device even with a host copy
  int or_windows = -1;  // OR: 1 = window-parallel kernel, 0 = candidate-driven kernel, -1 = auto
  int bound_slack_ppm = 0;  // block-max bounds are widened by (1 + ppm * 1e-6), see block_max_score
                        // (windows for exhaustive scans, candidates when pruning)
  // unpruned unions, doc-major (tq_xunion.hip): queries whose lists together hold at least
  // max_doc / xunion_ratio postings (0 = never), if the batch has at least xunion_min_queries of them
  int xunion_ratio = 64;
  int xunion_min_queries = 64;
  // tq_submit / tq_search_one: how long the leader of a batch waits for the callers of the previous
  // batch to come back with their next query (0 = launch with whatever is pending)
  int submit_window_us = 100;
};

}  // namespace

// The big per-batch buffers — partial / result lists, the staging lists of the two term-major launches —
// exist once per DEVICE, not once per segment: a device runs one batch at a time anyway (the kernels fill
// it), and 100 segments on a GPU must not mean 100 copies (8 segments held 13.9 GB of scratch in round 3).
// A batch takes the lock when it sizes the buffers and keeps it until the event behind its last kernel is
// recorded; a batch on another stream than the previous user's first waits for that event (stream side).
struct DeviceScratch {
  std::mutex m;
  DevBuf partials, share_stage, ashare_stage;
  hipEvent_t ev_last = nullptr;
  hipStream_t last_stream = nullptr;
  bool in_flight = false;
};
struct tq_ctx {
  std::vector<int> devices;
  std::mutex m;
  std::map<int, DeviceScratch *> scratch;
  DeviceScratch *scratch_for(int device) {
    std::lock_guard<std::mutex> lk(m);
    DeviceScratch *&p = scratch[device];
    if (!p) p = new DeviceScratch();
    return p;
  }
  ~tq_ctx() {
    for (auto &kv : scratch) {
      (void)hipSetDevice(kv.first);
      if (kv.second->ev_last) (void)hipEventDestroy(kv.second->ev_last);
      kv.second->partials.release();
      kv.second->share_stage.release();
      kv.second->ashare_stage.release();
      delete kv.second;
    }
  }
};

int tq_internal_fail(int code, const char *where, const char *what) {
  return fail(code, "%s: %s", where, what);
}
bool tq_internal_ctx_has_device(const tq_ctx *ctx, int device) {
  return std::find(ctx->devices.begin(), ctx->devices.end(), device) != ctx->devices.end();
}

struct tq_segment {
  tq_ctx *ctx = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  uint32_t max_doc = 0;
  uint8_t record_option = 0;
  std::vector<uint8_t> h_idx, h_pos;  // host copies (empty for a device-resident upload)
  size_t idx_len = 0, pos_len = 0;    // sizes of the sub-files in HBM
  TqpInfo *d_tp_info = nullptr;       // device-side prepare: result slots (info + positions result)
  uint8_t *d_idx = nullptr, *d_pos = nullptr, *d_fn = nullptr, *d_alive = nullptr;
  uint64_t *d_docmat = nullptr;  // doc-major matrix of the dense lists (TqdSegment::docmat)
  uint32_t n_mat_slots = 0;
  TqdSegment dseg{};
  std::vector<TermHost> terms;
  std::vector<TqdTerm> h_dterms;
  TqdTerm *d_terms = nullptr;
  size_t d_terms_cap = 0;
  bool d_terms_dirty = false;
  size_t dense_bytes_total = 0;
  // The side tables of the dense lists (bitmaps + rank directories, byte-wide tfs, position directories,
  // plain lists) live in ONE device allocation of dense_budget() bytes, made with the first of them:
  // the term-major launches address them as 32-bit offsets (8-byte units) from its base.  (With one
  // hipMalloc per table the allocator now and then returned addresses more than 32 GB apart and the
  // launches silently fell back to the per-query kernels.)  What does not fit falls back to hipMalloc.
  uint8_t *dense_arena = nullptr;
  size_t dense_arena_cap = 0, dense_arena_used = 0;
  std::vector<void *> dense_extra;  // tables allocated outside the arena
  // resident bytes by kind (tq_segment_get_stats)
  size_t bytes_term_tables = 0, bytes_bitmaps = 0, bytes_docmat = 0, bytes_posdir = 0, bytes_alive = 0;
  uint32_t n_dense_lists = 0;
  std::unordered_map<uint64_t, uint32_t> term_by_off;
  // lists named by tq_segment_reserve_columns (postings_off): only they get doc-matrix columns
  std::unordered_map<uint64_t, bool> reserved_cols;
  bool cols_reserved = false;
  // batch scratch
  DevBuf d_stage, d_out_scores, d_out_docs, d_out_counts, d_misc, d_thr, d_qmatches;
  DevBuf d_share_words;   // shared-union launch: per-query words
  DevBuf d_ashare_words;  // shared-intersection launch (runs next to the shared-union one)
  DeviceScratch *dscratch = nullptr;  // partial / result lists and staging lists: the device's (tq_ctx)
  // the shared-union launch addresses bitmaps / byte-wide tfs as 32-bit offsets (8-byte units) from
  // the lowest such table: usable while all of them lie within 32 GB of device addresses
  size_t share_span_terms = 0;  // number of terms the span was computed over
  uint64_t share_table_lo = 0;
  bool share_span_ok = true;
  uint32_t last_batch_queries = 0;
  PinnedBuf h_stage, h_out;
  // timing: a ring of event quadruples, one per batch, so that pipelined batches (no host sync
  // between them) can all be timed; tq_last_batch_stats averages the batches since its last call
  static constexpr int kTimingRing = 16;
  hipEvent_t ev_stage_done = nullptr, ev_fork = nullptr, ev_join = nullptr;
  // The per-segment scratch (query descriptors, partial lists, threshold slots, counters, the
  // side stream) is shared by consecutive batches: work enqueued on another stream than the
  // previous batch's must first wait for that batch (ev_batch_done, recorded at its end).
  hipEvent_t ev_batch_done = nullptr;
  hipStream_t last_stream = nullptr;
  bool batch_in_flight = false;
  hipStream_t side_stream = nullptr;  // the launch groups of one batch run concurrently
  // The batch's staging blob goes up on a stream of its own, into one of two device buffers, while
  // the previous batch's kernels still run (TQ_COPY_STREAM=0: on the batch's stream, one buffer).
  // Measured with SDMA copies: step 5.19 -> 5.12 ms on 60-step runs and a steadier step time;
  // round 1's attempt (one buffer, blit copies) had lost 8 %
  hipStream_t copy_stream = nullptr;
  DevBuf d_stage_alt;                        // the second staging buffer (d_stage is the first)
  hipEvent_t ev_copy_done[2] = {nullptr, nullptr}, ev_buf_free[2] = {nullptr, nullptr};
  bool buf_used[2] = {false, false};
  uint64_t batches_enqueued = 0;
  hipEvent_t ev_t0[kTimingRing] = {}, ev_t1[kTimingRing] = {}, ev_k0[kTimingRing] = {},
             ev_k1[kTimingRing] = {};
  uint64_t batches_timed = 0, batches_reported = 0;
  bool stage_in_flight = false;
  bool thr_seeded = false;  // (TQ_KEEP_THR experiments: the slots were zeroed once)
  double host_ms_sum = 0;   // host time inside tq_search_batch_device since the last stats call
  uint32_t host_ms_n = 0;
  unsigned long long *d_match_counter = nullptr;
  Options opt;
  size_t dense_budget() const { return (size_t)opt.dense_budget_x * (idx_len + pos_len + max_doc); }
  bool device_prepare() const { return h_idx.empty() || opt.device_prepare != 0; }
  tq_batch_stats stats{};
  bool stats_pending = false;
  // host planner scratch (launch groups, chunk tables): kept between batches so that planning a
  // batch does not start by page-faulting tens of megabytes of fresh vectors
  struct PlanScratch *plan = nullptr;
  // One call at a time works on a segment's state (term table, planner scratch, staging buffers):
  // every entry point takes this lock, so concurrent callers are serialised, not undefined.
  // (recursive: tq_count_batch -> tq_search_batch -> ...)
  std::recursive_mutex exec_m;
  // tq_submit / tq_wait / tq_search_one: single queries of concurrent callers, coalesced into batches
  struct SubmitQueue *submit = nullptr;
};

void tq_free_plan_scratch(PlanScratch *p);  // (defined next to the planner)
void tq_free_submit_queue(struct SubmitQueue *q);
#define TQ_SEGMENT_LOCK(seg) std::lock_guard<std::recursive_mutex> tq_exec_lock_((seg)->exec_m)

namespace {
int sync_terms(tq_segment *s, hipStream_t st);
int term_prepare_device(tq_segment *s, uint64_t postings_off, uint32_t postings_len,
                        uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                        tq_term_handle *out);
int build_dense_device(tq_segment *s, uint32_t handle);
int register_term(tq_segment *s, const TqdTerm &dt, const TermHost &th, uint64_t postings_off,
                  tq_term_handle *out);
int add_to_doc_signatures(tq_segment *s, uint32_t handle);
int ensure_docmat(tq_segment *s);

// Dense lists also get their term freqs as one byte per posting (255 = "255 or more: read the
// packed value"): with the posting index from the bitmap's rank the tf of a candidate is ONE load,
// where block record -> packed tf bits are two dependent ones (the shared-union kernel's scoring
// stage is a chain of dependent gathers, 1.6 us each under load).  d_tfs = the decoded tfs.
// a side table of a dense list: from the segment's arena, else a device allocation of its own
int dense_alloc(tq_segment *s, size_t bytes, void **out) {
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (!s->dense_arena && s->dense_arena_cap == 0) {
    const size_t cap = std::max<size_t>(s->dense_budget(), (size_t)1 << 20) + PAD;
    void *base = nullptr;
    if (hipMalloc(&base, cap) == hipSuccess) {
      s->dense_arena = (uint8_t *)base;
      s->dense_arena_cap = cap;
    } else {
      (void)hipGetLastError();
      s->dense_arena_cap = 1;  // (tried once: every table gets its own allocation)
    }
  }
  if (s->dense_arena && s->dense_arena_used + need + PAD <= s->dense_arena_cap) {
    *out = s->dense_arena + s->dense_arena_used;
    s->dense_arena_used += need;
    return TQ_OK;
  }
  void *ptr = nullptr;
  HIP_TRY(hipMalloc(&ptr, bytes));
  s->dense_extra.push_back(ptr);
  *out = ptr;
  return TQ_OK;
}
// (tables are only ever released with the segment; a failed build leaves its bytes unused)
void dense_release(tq_segment *s, void *ptr) {
  for (size_t i = 0; i < s->dense_extra.size(); ++i)
    if (s->dense_extra[i] == ptr) {
      (void)hipFree(ptr);
      s->dense_extra.erase(s->dense_extra.begin() + (long)i);
      return;
    }
}

int build_tf8(tq_segment *s, uint32_t handle, const uint32_t *d_tfs) {
  TermHost &t = s->terms[handle];
  const size_t bytes = ((size_t)t.doc_freq + 7) & ~(size_t)7;
  if (s->dense_bytes_total + bytes > s->dense_budget()) return TQ_OK;
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, bytes + PAD, &blob);
    if (arc != TQ_OK) return arc;
  }
  hipError_t e = tqk_launch_tf8_pack(d_tfs, t.doc_freq, (uint8_t *)blob, s->stream);
  if (e != hipSuccess) {
    dense_release(s, blob);
    return fail(TQ_ERR_HIP, "tf8 pack: %s", hipGetErrorString(e));
  }
  t.tf8_blob = blob;
  s->h_dterms[handle].tf8 = (const uint8_t *)blob;
  s->dense_bytes_total += bytes;
  s->bytes_bitmaps += bytes;
  s->d_terms_dirty = true;
  return TQ_OK;
}

// A list WITHOUT a bitmap as plain arrays — doc ids, then min(tf, 255) per posting — for the
// doc-major union launch (tq_xunion.hip), which scatters such a list into its tile row with a
// cursor instead of decoding blocks.  Built on the batch's stream the first time an unpruned
// union needs the list; counts against the side tables' budget (false = over it: *ok stays false).
int build_flat(tq_segment *s, uint32_t handle, hipStream_t st, bool *ok) {
  TermHost &t = s->terms[handle];
  *ok = t.flat_blob != nullptr;
  if (*ok || t.doc_freq == 0) return TQ_OK;
  const size_t doc_bytes = ((size_t)t.doc_freq * sizeof(uint32_t) + 15) & ~(size_t)15;
  const size_t bytes = doc_bytes + (((size_t)t.doc_freq + 15) & ~(size_t)15);
  if (s->dense_bytes_total + bytes > s->dense_budget()) return TQ_OK;
  int rc = sync_terms(s, st);
  if (rc != TQ_OK) return rc;
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, bytes + PAD, &blob);
    if (arc != TQ_OK) return arc;
  }
  const hipError_t e = tqk_launch_flat_list(s->dseg, s->d_terms, handle, t.n_blocks, (uint32_t *)blob,
                                            (uint8_t *)blob + doc_bytes, st);
  if (e != hipSuccess) {
    dense_release(s, blob);
    return fail(TQ_ERR_HIP, "flat list: %s", hipGetErrorString(e));
  }
  t.flat_blob = blob;
  s->dense_bytes_total += bytes;
  s->bytes_bitmaps += bytes;
  *ok = true;
  return TQ_OK;
}

// Orders work about to be enqueued on `st` after the segment's previous batch, whatever stream
// that batch ran on (no-op when it is the same stream: stream order already holds).
int order_after_last_batch(tq_segment *s, hipStream_t st) {
  if (s->batch_in_flight && s->last_stream != st)
    HIP_TRY(hipStreamWaitEvent(st, s->ev_batch_done, 0));
  return TQ_OK;
}
// Host-side wait for everything the segment has in flight (its own stream and the last batch).
int wait_segment_idle(tq_segment *s) {
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->batch_in_flight) {
    HIP_TRY(hipEventSynchronize(s->ev_batch_done));
    s->batch_in_flight = false;
  }
  return TQ_OK;
}

// The doc matrix (TqdSegment::docmat): allocated with the first list that needs it.
int ensure_docmat(tq_segment *s) {
  if (s->d_docmat) return TQ_OK;
  const size_t mat_bytes = (size_t)s->max_doc * sizeof(uint64_t);
  if (s->dense_bytes_total + mat_bytes > s->dense_budget()) return TQ_OK;  // (stays null: over budget)
  HIP_TRY(hipMalloc((void **)&s->d_docmat, mat_bytes + PAD));
  HIP_TRY(hipMemsetAsync((uint8_t *)s->d_docmat + mat_bytes, 0, PAD, s->stream));
  const hipError_t e = tqk_launch_docmat_init(s->d_docmat, s->d_fn, s->dseg.const_fieldnorm_id, s->max_doc, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat init: %s", hipGetErrorString(e));
  s->dense_bytes_total += mat_bytes;
  s->bytes_docmat = mat_bytes;
  s->dseg.docmat = s->d_docmat;
  return TQ_OK;
}

// Lists WITHOUT a column in the doc matrix (the sparse, high-weight lists; dense lists beyond the
// 40 columns) share the top 16 bits of the doc-matrix words: every such list sets bit
// 48 + hash(handle) of the docs it holds.  A clear bit proves "not in the list"; a set bit means
// "maybe" (another list with the same bit, or this one).  The union kernels test it where they
// used to assume the list holds every candidate — a rare list holds a fraction of a percent of
// them, and each wrong guess cost a seek and a block search.  The same gather that brings a
// candidate's fieldnorm id and column bits brings its signature.  Only prepared (queried) lists
// set bits; built by one decode of the list.
int add_to_doc_signatures(tq_segment *s, uint32_t handle) {
  if (!s->opt.docsig || !s->opt.docmat || !s->opt.dense || s->max_doc < 4096u) return TQ_OK;
  if ((s->h_dterms[handle].has_freq >> 8) & 0xFFu) return TQ_OK;  // the list has a column
  TermHost &t = s->terms[handle];
  if (t.doc_freq == 0) return TQ_OK;
  int rc = ensure_docmat(s);
  if (rc != TQ_OK) return rc;
  if (!s->d_docmat) return TQ_OK;
  rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  rc = s->d_misc.ensure(2 * bytes + 64);
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  hipError_t e = tqk_launch_decode_list(s->dseg, s->d_terms, handle, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  const uint32_t bit = (handle * 0x9E3779B1u) >> (32 - 4);  // 0 .. TQD_SIG_BITS - 1
  e = tqk_launch_docmat_set(s->d_docmat, dd, t.doc_freq, (TQD_SIG_SHIFT - 8u) + bit, s->max_doc, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat signature: %s", hipGetErrorString(e));
  s->h_dterms[handle].has_freq |= (bit + 1u) << 16;
  s->d_terms_dirty = true;
  return TQ_OK;
}

// Dense lists (doc_freq >= max_doc/TQD_DENSE_RATIO) also get a membership bitmap with a rank
// directory: the list is decoded once on the device (the same kernel as tq_decode_postings),
// and {32 doc bits, number of postings before them} pairs are uploaded.  A probe of doc d then
// costs one 8-byte load instead of a block decode; the posting index (=> block, slot, tf) falls
// out of the rank.  Derived data like the unrolled skip table; the index bytes stay untouched.
int build_dense(tq_segment *s, uint32_t handle) {
  int rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  TermHost &t = s->terms[handle];
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  rc = s->d_misc.ensure(2 * bytes + 64);
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  hipError_t e = tqk_launch_decode_list(s->dseg, s->d_terms, handle, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  rc = build_tf8(s, handle, dt);
  if (rc != TQ_OK) return rc;
  // the list's column of the doc matrix (first TQD_MAT_SLOTS dense lists of the segment, while
  // the matrix fits the same memory budget as the bitmaps)
  if (s->n_mat_slots < TQD_MAT_SLOTS && s->opt.docmat && t.wants_col) {
    {
      const int mrc = ensure_docmat(s);
      if (mrc != TQ_OK) return mrc;
    }
    if (s->d_docmat) {
      const uint32_t slot = s->n_mat_slots++;
      e = tqk_launch_docmat_set(s->d_docmat, dd, t.doc_freq, slot, s->max_doc, s->stream);
      if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat set: %s", hipGetErrorString(e));
      s->h_dterms[handle].has_freq |= (slot + 1u) << 8;
    }
  }
  std::vector<uint32_t> docs(t.doc_freq), tfs;
  HIP_TRY(hipMemcpyAsync(docs.data(), dd, bytes, hipMemcpyDeviceToHost, s->stream));
  const bool want_dir = t.positions_len > 0;
  if (want_dir) {
    tfs.resize(t.doc_freq);
    HIP_TRY(hipMemcpyAsync(tfs.data(), dt, bytes, hipMemcpyDeviceToHost, s->stream));
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  const size_t n_words = ((size_t)s->max_doc + 31) / 32 + 1;
  std::vector<uint2> tab(n_words, make_uint2(0u, 0u));
  uint32_t prev = 0;
  for (uint32_t i = 0; i < t.doc_freq; ++i) {
    const uint32_t d = docs[i];
    if (d >= s->max_doc || (i && d <= prev))
      return fail(TQ_ERR_FORMAT, "posting list not strictly increasing below max_doc");
    tab[d >> 5].x |= 1u << (d & 31u);
    prev = d;
  }
  uint32_t running = 0;
  for (size_t w = 0; w < n_words; ++w) {
    tab[w].y = running;
    running += (uint32_t)__builtin_popcount(tab[w].x);
  }
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, n_words * sizeof(uint2), &blob);
    if (arc != TQ_OK) return arc;
  }
  s->bytes_bitmaps += n_words * sizeof(uint2);
  ++s->n_dense_lists;
  hipError_t ce = hipMemcpy(blob, tab.data(), n_words * sizeof(uint2), hipMemcpyHostToDevice);
  if (ce != hipSuccess) {
    dense_release(s, blob);
    return fail(TQ_ERR_HIP, "dense upload: %s", hipGetErrorString(ce));
  }
  t.dense_blob = blob;
  s->h_dterms[handle].dense = (const uint2 *)blob;
  s->d_terms_dirty = true;
  if (want_dir) {  // position directory: positions before every fourth posting
    const size_t n_dir = ((size_t)t.doc_freq + 3) / 4 + 1;
    std::vector<uint32_t> dir(n_dir);
    uint64_t run = 0;
    for (uint32_t i = 0; i < t.doc_freq; ++i) {
      if ((i & 3u) == 0u) dir[i >> 2] = (uint32_t)run;
      run += tfs[i];
    }
    dir[n_dir - 1] = (uint32_t)run;
    if (run != t.n_positions)
      return fail(TQ_ERR_FORMAT, "term freqs sum to %llu positions, the stream holds %llu",
                  (unsigned long long)run, (unsigned long long)t.n_positions);
    void *db = nullptr;
    {
      const int arc = dense_alloc(s, n_dir * sizeof(uint32_t) + PAD, &db);
      if (arc != TQ_OK) return arc;
    }
    hipError_t de = hipMemcpy(db, dir.data(), n_dir * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (de != hipSuccess) {
      dense_release(s, db);
      return fail(TQ_ERR_HIP, "position directory upload: %s", hipGetErrorString(de));
    }
    t.posdir_blob = db;
    s->h_dterms[handle].pos_dir = (const uint32_t *)db;
    s->dense_bytes_total += n_dir * sizeof(uint32_t);
    s->bytes_posdir += n_dir * sizeof(uint32_t);
  }
  return TQ_OK;
}
}  // namespace

extern "C" {

const char *tq_last_error(void) { return g_last_error.c_str(); }

int tq_init(const int *device_ids, int n_devices, tq_ctx **out) {
  if (!out) return fail(TQ_ERR_INVALID, "tq_init: out is null");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(TQ_ERR_NO_DEVICE, "tq_init: no HIP device visible (%s)",
                e == hipSuccess ? "count=0" : hipGetErrorString(e));
  tq_ctx *ctx = new tq_ctx();
  if (!device_ids || n_devices <= 0) {
    ctx->devices.push_back(0);
  } else {
    for (int i = 0; i < n_devices; ++i) {
      if (device_ids[i] < 0 || device_ids[i] >= count) {
        delete ctx;
        return fail(TQ_ERR_INVALID, "tq_init: device %d out of range (%d visible)", device_ids[i],
                    count);
      }
      ctx->devices.push_back(device_ids[i]);
    }
  }
  *out = ctx;
  return TQ_OK;
}

void tq_shutdown(tq_ctx *ctx) { delete ctx; }

static int segment_upload_common(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *idx,
                                 size_t idx_len, const uint8_t *pos, size_t pos_len,
                                 const uint8_t *fieldnorm, size_t fn_len, uint8_t record_option,
                                 bool from_device, tq_segment **out) {
  if (!ctx || !out || !idx) return fail(TQ_ERR_INVALID, "tq_segment_upload: null argument");
  if (idx_len < 8) return fail(TQ_ERR_FORMAT, "idx sub-file shorter than its 8-byte header");
  if (record_option > TQ_WITH_FREQS_AND_POSITIONS)
    return fail(TQ_ERR_INVALID, "bad record_option %u", record_option);
  if (fieldnorm && fn_len < max_doc)
    return fail(TQ_ERR_FORMAT, "fieldnorm file has %zu bytes for max_doc %u", fn_len, max_doc);
  if (std::find(ctx->devices.begin(), ctx->devices.end(), device) == ctx->devices.end())
    return fail(TQ_ERR_INVALID, "device %d not part of this context", device);
  HIP_TRY(hipSetDevice(device));
  tq_segment *s = new tq_segment();
  s->ctx = ctx;
  s->device = device;
  s->dscratch = ctx->scratch_for(device);
  s->max_doc = max_doc;
  s->record_option = record_option;
  s->idx_len = idx_len;
  s->pos_len = (pos && pos_len) ? pos_len : 0;
  if (!from_device) {
    s->h_idx.assign(idx, idx + idx_len);
    if (pos && pos_len) s->h_pos.assign(pos, pos + pos_len);
  }
  const hipMemcpyKind kind = from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  auto up = [&](uint8_t **dst, const uint8_t *src, size_t n) -> int {
    HIP_TRY(hipMalloc((void **)dst, n + PAD));
    HIP_TRY(hipMemset(*dst + n, 0, PAD));
    HIP_TRY(hipMemcpy(*dst, src, n, kind));
    return TQ_OK;
  };
  int rc = up(&s->d_idx, idx, idx_len);
  if (rc == TQ_OK && pos && pos_len) rc = up(&s->d_pos, pos, pos_len);
  if (rc == TQ_OK && fieldnorm) rc = up(&s->d_fn, fieldnorm, max_doc);
  if (rc == TQ_OK) {
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_stage_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_batch_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipEventCreateWithFlags(&s->ev_copy_done[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_buf_free[i], hipEventDisableTiming);
    }
    for (int i = 0; i < tq_segment::kTimingRing; ++i) {
      if (e == hipSuccess) e = hipEventCreate(&s->ev_t0[i]);
      if (e == hipSuccess) e = hipEventCreate(&s->ev_t1[i]);
      if (e == hipSuccess) e = hipEventCreate(&s->ev_k0[i]);
      if (e == hipSuccess) e = hipEventCreate(&s->ev_k1[i]);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_match_counter, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_tp_info, 2 * sizeof(TqpInfo));
    if (e != hipSuccess) rc = fail(TQ_ERR_HIP, "segment setup: %s", hipGetErrorString(e));
  }
  if (rc != TQ_OK) {
    tq_segment_free(s);
    return rc;
  }
  s->dseg.idx = s->d_idx;
  s->dseg.pos = s->d_pos;
  s->dseg.fieldnorm = s->d_fn;
  s->dseg.max_doc = max_doc;
  s->dseg.const_fieldnorm_id = 1;  // FieldNormReader::constant(max_doc, 1)
  s->dseg.min_fieldnorm_id = 1;
  if (fieldnorm && !from_device) {
    uint8_t mn = 255;
    for (uint32_t d = 0; d < max_doc; ++d) mn = fieldnorm[d] < mn ? fieldnorm[d] : mn;
    s->dseg.min_fieldnorm_id = max_doc ? mn : 0;
  } else if (fieldnorm) {  // smallest fieldnorm id present: a device reduction, 4 bytes back
    uint32_t *slot = (uint32_t *)s->d_tp_info;
    uint32_t mn = 255;
    hipError_t e = hipMemcpy(slot, &mn, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = tqp_launch_min_fieldnorm(s->d_fn, max_doc, slot, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&mn, slot, 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
      tq_segment_free(s);
      return fail(TQ_ERR_HIP, "fieldnorm scan: %s", hipGetErrorString(e));
    }
    s->dseg.min_fieldnorm_id = max_doc ? mn : 0;
  }
  *out = s;
  return TQ_OK;
}

int tq_segment_upload(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *idx,
                      size_t idx_len, const uint8_t *pos, size_t pos_len, const uint8_t *fieldnorm,
                      size_t fn_len, uint8_t record_option, tq_segment **out) {
  return segment_upload_common(ctx, device, max_doc, idx, idx_len, pos, pos_len, fieldnorm, fn_len,
                               record_option, false, out);
}

int tq_segment_upload_device(tq_ctx *ctx, int device, uint32_t max_doc, const uint8_t *d_idx,
                             size_t idx_len, const uint8_t *d_pos, size_t pos_len,
                             const uint8_t *d_fieldnorm, size_t fn_len, uint8_t record_option,
                             tq_segment **out) {
  return segment_upload_common(ctx, device, max_doc, d_idx, idx_len, d_pos, pos_len, d_fieldnorm,
                               fn_len, record_option, true, out);
}

void tq_segment_free(tq_segment *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->batch_in_flight && s->ev_batch_done) (void)hipEventSynchronize(s->ev_batch_done);
  for (auto &t : s->terms)
    if (t.blob) (void)hipFree(t.blob);
  for (auto &t : s->terms)
    if (t.pos_blob) (void)hipFree(t.pos_blob);
  // (bitmaps, byte-wide tfs, position directories, plain lists: the arena and its overflow)
  if (s->dense_arena) (void)hipFree(s->dense_arena);
  for (void *ptr : s->dense_extra) (void)hipFree(ptr);
  if (s->d_terms) (void)hipFree(s->d_terms);
  if (s->d_idx) (void)hipFree(s->d_idx);
  if (s->d_pos) (void)hipFree(s->d_pos);
  if (s->d_fn) (void)hipFree(s->d_fn);
  if (s->d_alive) (void)hipFree(s->d_alive);
  if (s->d_docmat) (void)hipFree(s->d_docmat);
  if (s->d_tp_info) (void)hipFree(s->d_tp_info);
  if (s->d_match_counter) (void)hipFree(s->d_match_counter);
  s->d_stage.release();
  s->d_out_scores.release();
  s->d_out_docs.release();
  s->d_out_counts.release();
  s->d_misc.release();
  s->d_thr.release();
  tq_free_plan_scratch(s->plan);
  s->plan = nullptr;
  tq_free_submit_queue(s->submit);
  s->submit = nullptr;
  s->d_qmatches.release();
  s->d_share_words.release();
  s->d_ashare_words.release();
  s->h_stage.release();
  s->h_out.release();
  if (s->side_stream) (void)hipStreamSynchronize(s->side_stream);
  if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
  for (hipEvent_t ev : {s->ev_stage_done, s->ev_fork, s->ev_join, s->ev_batch_done, s->ev_copy_done[0],
                        s->ev_copy_done[1], s->ev_buf_free[0], s->ev_buf_free[1]})
    if (ev) (void)hipEventDestroy(ev);
  if (s->side_stream) (void)hipStreamDestroy(s->side_stream);
  if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
  s->d_stage_alt.release();
  for (int i = 0; i < tq_segment::kTimingRing; ++i)
    for (hipEvent_t ev : {s->ev_t0[i], s->ev_t1[i], s->ev_k0[i], s->ev_k1[i]})
      if (ev) (void)hipEventDestroy(ev);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

int tq_term_prepare(tq_segment *s, uint64_t postings_off, uint32_t postings_len,
                    uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                    tq_term_handle *out) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_term_prepare: null argument");
  TQ_SEGMENT_LOCK(s);
  auto it = s->term_by_off.find(postings_off);
  if (it != s->term_by_off.end()) {
    *out = it->second;
    return TQ_OK;
  }
  if (doc_freq == 0) return fail(TQ_ERR_INVALID, "tq_term_prepare: doc_freq 0 (term absent)");
  const size_t body_len = s->idx_len - 8;
  if (postings_off > body_len || (uint64_t)postings_len > body_len - postings_off)
    return fail(TQ_ERR_FORMAT, "postings_range [%llu,+%u) outside the idx body (%zu)",
                (unsigned long long)postings_off, postings_len, body_len);
  HIP_TRY(hipSetDevice(s->device));
  if (s->device_prepare())
    return term_prepare_device(s, postings_off, postings_len, positions_off, positions_len, doc_freq,
                               out);
  const uint8_t *data = s->h_idx.data() + 8 + postings_off;
  const size_t len = postings_len;
  const uint64_t abs0 = 8 + postings_off;  // offset of `data` inside the uploaded sub-file

  int record = s->record_option;
  const uint32_t n_full = doc_freq / 128u, n_tail = doc_freq % 128u;
  size_t at = 0;
  const uint8_t *skip = nullptr;
  size_t skip_len = 0;
  if (doc_freq >= 128u) {  // block_segment_postings.rs:78-88
    uint64_t sl;
    if (!read_vint(data, len, at, sl) || sl > len - at)
      return fail(TQ_ERR_FORMAT, "bad skip_len for term at %llu", (unsigned long long)postings_off);
    skip = data + at;
    skip_len = (size_t)sl;
    at += skip_len;
    if (skip_len < 8ull * n_full) record = TQ_BASIC;  // :107-116 (JSON terms without freqs)
  }
  const size_t entry = record == TQ_BASIC ? 5 : (record == TQ_WITH_FREQS ? 8 : 12);
  if (skip_len < entry * n_full)
    return fail(TQ_ERR_FORMAT, "skip data too short: %zu < %zu", skip_len, entry * n_full);
  const bool has_freq = record != TQ_BASIC;
  const size_t payload = at;

  const uint32_t n_blocks = n_full + (n_tail ? 1u : 0u);
  std::vector<uint32_t> b_last(n_blocks), b_meta(n_blocks), b_off(n_blocks);
  std::vector<uint32_t> block_pos(n_blocks + 1, 0);
  size_t running = 0;
  uint64_t running_pos = 0;
  uint32_t last_doc = 0;
  for (uint32_t i = 0; i < n_full; ++i) {  // skip.rs:205-253,275-302
    const uint8_t *e = skip + entry * i;
    const uint32_t ld = rd32(e);
    const uint32_t doc_bits = e[4] & 0x1Fu, strict = (e[4] >> 6) & 1u;
    uint32_t tf_bits = 0, tf_sum = 0, bm_fn = 0, bm_tf = 0;
    if (record == TQ_WITH_FREQS) {
      tf_bits = e[5];
      bm_fn = e[6];
      bm_tf = e[7];
    } else if (record == TQ_WITH_FREQS_AND_POSITIONS) {
      tf_bits = e[5];
      tf_sum = rd32(e + 6);
      bm_fn = e[10];
      bm_tf = e[11];
    }
    if (tf_bits > 32u) return fail(TQ_ERR_FORMAT, "tf bit width %u > 32", tf_bits);
    if (i && ld <= last_doc) return fail(TQ_ERR_FORMAT, "skip last_doc not increasing");
    if (running_pos > 0xFFFFFFFFull)
      return fail(TQ_ERR_UNSUPPORTED, "term with more than 2^32 positions");
    b_last[i] = ld;
    b_meta[i] = doc_bits | (strict << 6) | (tf_bits << 8) | (bm_fn << 16) | (bm_tf << 24);
    b_off[i] = (uint32_t)running;  // < postings_len, a u32 (term_info.rs:10-17)
    block_pos[i] = (uint32_t)running_pos;
    running += 16u * (size_t)(doc_bits + tf_bits);
    running_pos += tf_sum;
    last_doc = ld;
  }
  if (payload + running > len) return fail(TQ_ERR_FORMAT, "bitpacked payload exceeds the list");
  std::vector<uint32_t> tail_docs(n_tail), tail_tfs(n_tail, 1u);
  if (n_tail) {  // vint.rs:44-108; docs delta from the last full block (0 if none)
    size_t t = payload + running;
    uint32_t prev = n_full ? last_doc : 0u;
    for (uint32_t i = 0; i < n_tail; ++i) {
      uint32_t d;
      if (!read_vint32_block(data, len, t, d)) return fail(TQ_ERR_FORMAT, "truncated vint docs");
      prev += d;
      tail_docs[i] = prev;
    }
    if (has_freq && t < len) {
      for (uint32_t i = 0; i < n_tail; ++i)
        if (!read_vint32_block(data, len, t, tail_tfs[i]))
          return fail(TQ_ERR_FORMAT, "truncated vint term freqs");
    }
    if (running_pos > 0xFFFFFFFFull)
      return fail(TQ_ERR_UNSUPPORTED, "term with more than 2^32 positions");
    b_last[n_full] = tail_docs[n_tail - 1];
    b_meta[n_full] = 0xFFFFFFFFu;
    b_off[n_full] = 0;
    block_pos[n_full] = (uint32_t)running_pos;
    if (record == TQ_WITH_FREQS_AND_POSITIONS)  // tf sums only index a positions stream
      for (uint32_t i = 0; i < n_tail; ++i) running_pos += tail_tfs[i];
    last_doc = tail_docs[n_tail - 1];
  }
  if (running_pos > 0xFFFFFFFFull)
    return fail(TQ_ERR_UNSUPPORTED, "term with more than 2^32 positions");
  block_pos[n_blocks] = (uint32_t)running_pos;
  if (last_doc >= TQ_TERMINATED) return fail(TQ_ERR_FORMAT, "doc id >= TERMINATED");
  if (last_doc >= s->max_doc)
    return fail(TQ_ERR_FORMAT, "doc id %u >= max_doc %u", last_doc, s->max_doc);

  // coarse[b] = first block j with last_doc[j] >= b << shift, about one block per bucket
  uint32_t shift = 7;
  while (shift < 31 && ((uint64_t)(s->max_doc - 1) >> shift) + 1 > 2ull * n_blocks + 2) ++shift;
  const uint32_t n_buckets = (uint32_t)(((uint64_t)(s->max_doc - 1)) >> shift) + 1;
  std::vector<uint32_t> coarse(n_buckets + 1);
  {
    uint32_t j = 0;
    for (uint32_t b = 0; b <= n_buckets; ++b) {
      const uint64_t lo = (uint64_t)b << shift;
      while (j < n_blocks && (uint64_t)b_last[j] < lo) ++j;
      coarse[b] = j;
    }
  }

  // positions stream (positions/reader.rs:43-56,84-101)
  std::vector<uint64_t> pos_block_off;
  std::vector<uint8_t> pos_widths;
  std::vector<uint32_t> pos_tail;
  const bool want_pos = s->record_option == TQ_WITH_FREQS_AND_POSITIONS && !s->h_pos.empty() &&
                        record == TQ_WITH_FREQS_AND_POSITIONS;
  if (want_pos) {
    if (positions_off > s->h_pos.size() || (uint64_t)positions_len > s->h_pos.size() - positions_off)
      return fail(TQ_ERR_FORMAT, "positions_range outside the pos file");
    const uint8_t *pd = s->h_pos.data() + positions_off;
    size_t pa = 0;
    uint64_t nb;
    if (!read_vint(pd, positions_len, pa, nb) || nb > positions_len - pa)
      return fail(TQ_ERR_FORMAT, "bad positions header");
    pos_widths.assign(pd + pa, pd + pa + nb);
    pa += (size_t)nb;
    size_t prun = 0;
    pos_block_off.resize((size_t)nb);
    for (size_t i = 0; i < nb; ++i) {
      if (pos_widths[i] > 32) return fail(TQ_ERR_FORMAT, "position bit width > 32");
      pos_block_off[i] = (uint64_t)(positions_off + pa + prun) | ((uint64_t)pos_widths[i] << 56);
      prun += 16u * (size_t)pos_widths[i];
    }
    size_t t = pa + prun;
    if (t > positions_len) return fail(TQ_ERR_FORMAT, "bitpacked positions exceed the range");
    while (t < positions_len) {  // uncompress_vint_unsorted_until_end
      uint32_t v;
      if (!read_vint32_block(pd, positions_len, t, v))
        return fail(TQ_ERR_FORMAT, "truncated vint positions");
      pos_tail.push_back(v);
    }
    const uint64_t n_pos = (uint64_t)nb * 128u + pos_tail.size();
    if (n_pos != running_pos)
      return fail(TQ_ERR_FORMAT, "positions stream holds %llu values, postings say %llu",
                  (unsigned long long)n_pos, (unsigned long long)running_pos);
  }

  // one blob holding every per-term array
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t total = 0;
  auto place = [&](size_t bytes) {
    const size_t o = total;
    total = align16(total + bytes);
    return o;
  };
  const size_t o_rec = place(16 * (size_t)(n_blocks + 1));
  const size_t o_coarse = place(4 * coarse.size());
  const size_t o_tdocs = place(4 * (size_t)n_tail);
  const size_t o_ttfs = place(4 * (size_t)n_tail);
  const size_t o_pboff = place(8 * pos_block_off.size());
  const size_t o_ptail = place(4 * pos_tail.size());
  total += PAD;
  std::vector<uint8_t> hb(total, 0);
  for (uint32_t i = 0; i <= n_blocks; ++i) {
    const uint32_t r[4] = {i < n_blocks ? b_last[i] : TQ_TERMINATED, i < n_blocks ? b_meta[i] : 0u,
                           i < n_blocks ? b_off[i] : 0u, block_pos[i]};
    memcpy(hb.data() + o_rec + 16 * (size_t)i, r, 16);
  }
  memcpy(hb.data() + o_coarse, coarse.data(), 4 * coarse.size());
  if (n_tail) {
    memcpy(hb.data() + o_tdocs, tail_docs.data(), 4 * (size_t)n_tail);
    memcpy(hb.data() + o_ttfs, tail_tfs.data(), 4 * (size_t)n_tail);
  }
  if (!pos_block_off.empty()) memcpy(hb.data() + o_pboff, pos_block_off.data(), 8 * pos_block_off.size());
  if (!pos_tail.empty()) memcpy(hb.data() + o_ptail, pos_tail.data(), 4 * pos_tail.size());
  uint8_t *blob = nullptr;
  HIP_TRY(hipMalloc((void **)&blob, total));
  s->bytes_term_tables += total;
  hipError_t ce = hipMemcpy(blob, hb.data(), total, hipMemcpyHostToDevice);
  if (ce != hipSuccess) {
    (void)hipFree(blob);
    return fail(TQ_ERR_HIP, "term upload: %s", hipGetErrorString(ce));
  }
  TqdTerm dt{};
  dt.rec = (const uint4 *)(blob + o_rec);
  dt.coarse = (const uint32_t *)(blob + o_coarse);
  dt.tail_docs = (const uint32_t *)(blob + o_tdocs);
  dt.tail_tfs = (const uint32_t *)(blob + o_ttfs);
  dt.pos_blk = (const uint64_t *)(blob + o_pboff);
  dt.pos_tail = (const uint32_t *)(blob + o_ptail);
  dt.payload_base = abs0 + payload;
  dt.n_full = n_full;
  dt.n_tail = n_tail;
  dt.n_blocks = n_blocks;
  dt.doc_freq = doc_freq;
  dt.n_pos_blocks = (uint32_t)pos_block_off.size();
  dt.n_pos_tail = (uint32_t)pos_tail.size();
  dt.has_freq = has_freq ? 1u : 0u;
  dt.coarse_shift = shift;

  TermHost th;
  th.blob = blob;
  th.doc_freq = doc_freq;
  th.n_blocks = n_blocks;
  th.n_full = n_full;
  th.n_tail = n_tail;
  th.last_doc = last_doc;
  th.postings_len = postings_len;
  th.positions_len = want_pos ? positions_len : 0;
  th.n_positions = want_pos ? running_pos : 0;
  return register_term(s, dt, th, postings_off, out);
}

}  // extern "C"

namespace {

const char *tqp_message(uint32_t st) {
  switch (st) {
    case TQP_BAD_SKIP_LEN: return "bad skip_len";
    case TQP_SKIP_TOO_SHORT: return "skip data too short";
    case TQP_NOT_INCREASING: return "skip last_doc not increasing";
    case TQP_BAD_TF_WIDTH: return "tf bit width > 32";
    case TQP_TOO_MANY_POSITIONS: return "term with more than 2^32 positions";
    case TQP_PAYLOAD_TOO_LONG: return "bitpacked payload exceeds the list";
    case TQP_TRUNCATED_TAIL: return "truncated vint tail";
    case TQP_DOC_OUT_OF_RANGE: return "doc id >= max_doc / TERMINATED";
    case TQP_BAD_POS_HEADER: return "bad positions header";
    case TQP_POS_COUNT_MISMATCH: return "positions stream and postings disagree on the number of positions";
    case TQP_BAD_POS_WIDTH: return "position bit width > 32";
    case TQP_POS_PAYLOAD_TOO_LONG: return "bitpacked positions exceed the range";
    default: return "unknown";
  }
}

// the part of tq_term_prepare both paths share: the handle, and the dense-list structures
int register_term(tq_segment *s, const TqdTerm &dt, const TermHost &th, uint64_t postings_off,
                  tq_term_handle *out) {
  const uint32_t handle = (uint32_t)s->terms.size();
  s->terms.push_back(th);
  s->terms.back().wants_col = !s->cols_reserved || s->reserved_cols.count(postings_off) != 0;
  s->h_dterms.push_back(dt);
  s->d_terms_dirty = true;
  s->term_by_off.emplace(postings_off, handle);
  *out = handle;
  // 0.25 B/doc per bitmap: worth it for lists whose 128-doc blocks span few docs, and only while
  // the bitmaps together stay within a fixed multiple of the segment's own size
  const size_t dense_bytes = (((size_t)s->max_doc + 31) / 32 + 1) * sizeof(uint2);
  const size_t budget = s->dense_budget();
  if (s->opt.dense && s->max_doc >= 4096u &&
      (uint64_t)th.doc_freq * (uint64_t)s->opt.dense_ratio >= s->max_doc &&
      s->dense_bytes_total + dense_bytes <= budget) {
    s->dense_bytes_total += dense_bytes;
    const int rc = s->device_prepare() ? build_dense_device(s, handle) : build_dense(s, handle);
    if (rc != TQ_OK) return rc;
  }
  return add_to_doc_signatures(s, handle);
}

// tq_term_prepare without a host copy of the index: two small kernels walk the list's skip data
// and positions header where they lie in HBM (tq_prepare.hip); the host sizes the tables from
// TermInfo, and reads back 48 bytes of facts in between (no index bytes).
int term_prepare_device(tq_segment *s, uint64_t postings_off, uint32_t postings_len,
                        uint64_t positions_off, uint32_t positions_len, uint32_t doc_freq,
                        tq_term_handle *out) {
  const uint32_t n_full = doc_freq / 128u, n_tail = doc_freq % 128u;
  const uint32_t n_blocks = n_full + (n_tail ? 1u : 0u);
  uint32_t shift = 7;
  while (shift < 31 && ((uint64_t)(s->max_doc - 1) >> shift) + 1 > 2ull * n_blocks + 2) ++shift;
  const uint32_t n_buckets = (uint32_t)(((uint64_t)(s->max_doc - 1)) >> shift) + 1;
  const bool maybe_pos = s->record_option == TQ_WITH_FREQS_AND_POSITIONS && s->d_pos != nullptr;
  if (maybe_pos && (positions_off > s->pos_len || (uint64_t)positions_len > s->pos_len - positions_off))
    return fail(TQ_ERR_FORMAT, "positions_range outside the pos file");
  auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
  size_t total = 0;
  auto place = [&](size_t bytes) {
    const size_t o = total;
    total = align16(total + bytes);
    return o;
  };
  const size_t o_rec = place(16 * (size_t)(n_blocks + 1));
  const size_t o_coarse = place(4 * (size_t)(n_buckets + 1));
  const size_t o_tdocs = place(4 * (size_t)n_tail);
  const size_t o_ttfs = place(4 * (size_t)n_tail);
  total += PAD;
  uint8_t *blob = nullptr;
  HIP_TRY(hipMalloc((void **)&blob, total));
  s->bytes_term_tables += total;
  auto bail = [&](int rc) {
    (void)hipFree(blob);
    return rc;
  };
  hipError_t e = hipMemsetAsync(blob, 0, total, s->stream);
  TqpPostingsParams pp{};
  pp.idx = s->d_idx;
  pp.pos = s->d_pos;
  pp.postings_off = postings_off;
  pp.positions_off = positions_off;
  pp.postings_len = postings_len;
  pp.positions_len = positions_len;
  pp.doc_freq = doc_freq;
  pp.record_option = s->record_option;
  pp.max_doc = s->max_doc;
  pp.want_pos = maybe_pos ? 1u : 0u;
  pp.rec = (uint4 *)(blob + o_rec);
  pp.tail_docs = (uint32_t *)(blob + o_tdocs);
  pp.tail_tfs = (uint32_t *)(blob + o_ttfs);
  pp.info = s->d_tp_info;
  if (e == hipSuccess) e = tqp_launch_postings(pp, s->stream);
  TqpInfo info{};
  if (e == hipSuccess) e = hipMemcpyAsync(&info, s->d_tp_info, sizeof info, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess) return bail(fail(TQ_ERR_HIP, "device term prepare: %s", hipGetErrorString(e)));
  if (info.status != TQP_OK)
    return bail(fail(info.status == TQP_TOO_MANY_POSITIONS ? TQ_ERR_UNSUPPORTED : TQ_ERR_FORMAT,
                     "term at %llu: %s", (unsigned long long)postings_off, tqp_message(info.status)));
  e = tqp_launch_coarse((const uint4 *)(blob + o_rec), n_blocks, shift, n_buckets,
                        (uint32_t *)(blob + o_coarse), s->stream);
  if (e != hipSuccess) return bail(fail(TQ_ERR_HIP, "coarse table: %s", hipGetErrorString(e)));
  // positions tables, sized from the walk
  const bool want_pos = maybe_pos && info.record == TQ_WITH_FREQS_AND_POSITIONS;
  uint8_t *pblob = nullptr;
  uint32_t n_pos_tail = 0;
  size_t o_pboff = 0, o_ptail = 0;
  if (want_pos) {
    const uint64_t tail_cap = info.n_positions - info.n_pos_blocks * 128ull;
    if (tail_cap > 127ull)
      return bail(fail(TQ_ERR_FORMAT, "positions stream and postings disagree on the number of positions"));
    size_t ptotal = 0;
    o_pboff = 0;
    ptotal = align16(8 * (size_t)info.n_pos_blocks);
    o_ptail = ptotal;
    ptotal = align16(ptotal + 4 * (size_t)tail_cap) + PAD;
    e = hipMalloc((void **)&pblob, ptotal);
    if (e == hipSuccess) s->bytes_term_tables += ptotal;
    if (e == hipSuccess) e = hipMemsetAsync(pblob, 0, ptotal, s->stream);
    TqpPositionsParams qp{};
    qp.pos = s->d_pos;
    qp.positions_off = positions_off;
    qp.pos_hdr = info.pos_hdr;
    qp.n_pos_blocks = info.n_pos_blocks;
    qp.n_positions = info.n_positions;
    qp.positions_len = positions_len;
    qp.pos_tail_cap = (uint32_t)tail_cap;
    qp.pos_blk = (uint64_t *)(pblob + o_pboff);
    qp.pos_tail = (uint32_t *)(pblob + o_ptail);
    qp.result = (uint32_t *)(s->d_tp_info + 1);
    uint32_t res[2] = {0, 0};
    if (e == hipSuccess) e = tqp_launch_positions(qp, s->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(res, s->d_tp_info + 1, sizeof res, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess || res[0] != TQP_OK) {
      if (pblob) (void)hipFree(pblob);
      return bail(e != hipSuccess ? fail(TQ_ERR_HIP, "device positions prepare: %s", hipGetErrorString(e))
                                  : fail(TQ_ERR_FORMAT, "term at %llu: %s", (unsigned long long)postings_off,
                                         tqp_message(res[0])));
    }
    n_pos_tail = res[1];
  } else {
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
  TqdTerm dt{};
  dt.rec = (const uint4 *)(blob + o_rec);
  dt.coarse = (const uint32_t *)(blob + o_coarse);
  dt.tail_docs = (const uint32_t *)(blob + o_tdocs);
  dt.tail_tfs = (const uint32_t *)(blob + o_ttfs);
  dt.pos_blk = (const uint64_t *)(pblob ? pblob + o_pboff : blob + o_rec);
  dt.pos_tail = (const uint32_t *)(pblob ? pblob + o_ptail : blob + o_rec);
  dt.payload_base = 8 + postings_off + info.payload;
  dt.n_full = n_full;
  dt.n_tail = n_tail;
  dt.n_blocks = n_blocks;
  dt.doc_freq = doc_freq;
  dt.n_pos_blocks = want_pos ? (uint32_t)info.n_pos_blocks : 0u;
  dt.n_pos_tail = n_pos_tail;
  dt.has_freq = info.record != TQ_BASIC ? 1u : 0u;
  dt.coarse_shift = shift;
  TermHost th;
  th.blob = blob;
  th.pos_blob = pblob;
  th.doc_freq = doc_freq;
  th.n_blocks = n_blocks;
  th.n_full = n_full;
  th.n_tail = n_tail;
  th.last_doc = info.last_doc;
  th.postings_len = postings_len;
  th.positions_len = want_pos ? positions_len : 0;
  th.n_positions = want_pos ? info.n_positions : 0;
  return register_term(s, dt, th, postings_off, out);
}

// build_dense without the host: bitmap bits by atomic OR, rank directory and position directory
// by device scans; 4 bytes (the validity flag) come back.
int build_dense_device(tq_segment *s, uint32_t handle) {
  int rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  TermHost &t = s->terms[handle];
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  rc = s->d_misc.ensure(2 * bytes + 64);
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  hipError_t e = tqk_launch_decode_list(s->dseg, s->d_terms, handle, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  rc = build_tf8(s, handle, dt);
  if (rc != TQ_OK) return rc;
  const size_t n_words = ((size_t)s->max_doc + 31) / 32 + 1;
  void *blob = nullptr;
  {
    const int arc = dense_alloc(s, n_words * sizeof(uint2), &blob);
    if (arc != TQ_OK) return arc;
  }
  s->bytes_bitmaps += n_words * sizeof(uint2);
  ++s->n_dense_lists;
  uint32_t *bad = (uint32_t *)s->d_tp_info;
  e = hipMemsetAsync(blob, 0, n_words * sizeof(uint2), s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, s->stream);
  if (e == hipSuccess)
    e = tqp_launch_dense(dd, t.doc_freq, s->max_doc, (uint2 *)blob, (uint32_t)n_words, bad, s->stream);
  uint32_t h_bad = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || h_bad) {
    dense_release(s, blob);
    return e != hipSuccess ? fail(TQ_ERR_HIP, "dense tables: %s", hipGetErrorString(e))
                           : fail(TQ_ERR_FORMAT, "posting list not strictly increasing below max_doc");
  }
  t.dense_blob = blob;
  s->h_dterms[handle].dense = (const uint2 *)blob;
  s->d_terms_dirty = true;
  if (s->n_mat_slots < TQD_MAT_SLOTS && s->opt.docmat && t.wants_col) {  // the list's column of the doc matrix
    {
      const int mrc = ensure_docmat(s);
      if (mrc != TQ_OK) return mrc;
    }
    if (s->d_docmat) {
      const uint32_t slot = s->n_mat_slots++;
      e = tqk_launch_docmat_set(s->d_docmat, dd, t.doc_freq, slot, s->max_doc, s->stream);
      if (e != hipSuccess) return fail(TQ_ERR_HIP, "docmat set: %s", hipGetErrorString(e));
      s->h_dterms[handle].has_freq |= (slot + 1u) << 8;
    }
  }
  if (t.positions_len > 0) {  // position directory: positions before every fourth posting
    const size_t n_dir = ((size_t)t.doc_freq + 3) / 4 + 1;
    void *db = nullptr;
    {
      const int arc = dense_alloc(s, n_dir * sizeof(uint32_t) + PAD, &db);
      if (arc != TQ_OK) return arc;
    }
    e = tqp_launch_posdir(dt, t.doc_freq, (uint32_t *)db, (uint32_t)n_dir, s->stream);
    uint32_t total = 0;
    if (e == hipSuccess)
      e = hipMemcpyAsync(&total, (uint32_t *)db + (n_dir - 1), 4, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess || total != (uint32_t)t.n_positions) {
      dense_release(s, db);
      return e != hipSuccess ? fail(TQ_ERR_HIP, "position directory: %s", hipGetErrorString(e))
                             : fail(TQ_ERR_FORMAT, "term freqs sum to %u positions, the stream holds %llu",
                                    total, (unsigned long long)t.n_positions);
    }
    t.posdir_blob = db;
    s->h_dterms[handle].pos_dir = (const uint32_t *)db;
    s->dense_bytes_total += n_dir * sizeof(uint32_t);
    s->bytes_posdir += n_dir * sizeof(uint32_t);
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}

}  // namespace

namespace {

int sync_terms(tq_segment *s, hipStream_t st) {
  if (!s->d_terms_dirty) return TQ_OK;
  const size_t n = s->h_dterms.size();
  // nothing in flight may still read the table while it is rewritten (terms are prepared rarely)
  {
    const int wrc = wait_segment_idle(s);
    if (wrc != TQ_OK) return wrc;
  }
  if (n > s->d_terms_cap) {
    HIP_TRY(hipStreamSynchronize(st));
    if (s->d_terms) (void)hipFree(s->d_terms);
    s->d_terms = nullptr;
    size_t cap = std::max<size_t>(256, n * 2);
    HIP_TRY(hipMalloc((void **)&s->d_terms, cap * sizeof(TqdTerm)));
    s->d_terms_cap = cap;
  }
  HIP_TRY(hipMemcpy(s->d_terms, s->h_dterms.data(), n * sizeof(TqdTerm), hipMemcpyHostToDevice));
  s->d_terms_dirty = false;
  return TQ_OK;
}

// A grow-only array of plain structs whose resize() leaves new elements uninitialised (the
// descriptors of a 10 000-query batch are 3 MB: std::vector::resize would zero them just before
// they are overwritten).
template <typename T>
class PodVec {
 public:
  PodVec() = default;
  PodVec(const PodVec &) = delete;
  PodVec &operator=(const PodVec &) = delete;
  PodVec(PodVec &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr, o.n_ = o.cap_ = 0; }
  PodVec &operator=(PodVec &&o) noexcept {
    swap(o);
    return *this;
  }
  ~PodVec() { free(p_); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T *data() { return p_; }
  const T *data() const { return p_; }
  T &operator[](size_t i) { return p_[i]; }
  const T &operator[](size_t i) const { return p_[i]; }
  T &back() { return p_[n_ - 1]; }
  const T &back() const { return p_[n_ - 1]; }
  T *begin() { return p_; }
  T *end() { return p_ + n_; }
  const T *begin() const { return p_; }
  const T *end() const { return p_ + n_; }
  void clear() { n_ = 0; }
  void reserve(size_t n) {
    if (n <= cap_) return;
    const size_t ncap = std::max(n, cap_ * 2);
    T *np = (T *)malloc(ncap * sizeof(T));
    if (!np) throw std::bad_alloc();
    if (n_) memcpy(np, p_, n_ * sizeof(T));
    free(p_);
    p_ = np;
    cap_ = ncap;
  }
  void resize(size_t n) {  // (new elements are NOT initialised)
    reserve(n);
    n_ = n;
  }
  void push_back(const T &v) {
    if (n_ == cap_) reserve(n_ + 1);
    p_[n_++] = v;
  }
  void append(const T *first, const T *last) {
    const size_t n = (size_t)(last - first);
    reserve(n_ + n);
    if (n) memcpy(p_ + n_, first, n * sizeof(T));
    n_ += n;
  }
  void swap(PodVec &o) {
    std::swap(p_, o.p_);
    std::swap(n_, o.n_);
    std::swap(cap_, o.cap_);
  }

 private:
  T *p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

struct Group {
  int mode;
  PodVec<TqdQuery> queries;
  std::vector<uint32_t> out_index;
  std::vector<uint32_t> tile_starts;
  std::vector<uint4> chunk_recs;      // launch order: {first tile, end tile, first query, chunk}
  std::vector<uint32_t> tile_cost;  // per query, cost units per tile
  uint32_t total_tiles = 0, n_chunks = 0, max_k = 1;
  int kpl = 1;
  // offsets inside the staging blob
  size_t o_queries = 0, o_tiles = 0, o_outidx = 0, o_chunks = 0, o_perm = 0, o_sinks = 0;
  size_t o_leads = 0, o_tasks = 0;  // shared-union group
  void reset() {  // keeps the vectors' capacity
    queries.clear();
    out_index.clear();
    tile_starts.clear();
    chunk_recs.clear();
    tile_cost.clear();
    total_tiles = 0;
    n_chunks = 0;
    max_k = 1;
    kpl = 1;
    o_queries = o_tiles = o_outidx = o_chunks = o_perm = o_sinks = 0;
    o_leads = o_tasks = 0;
  }
};

}  // namespace

struct alignas(128) PlanSlab {  // chunk tables of one slab of queries (build_group_chunks); its own
                                // cache lines: the slabs' vector ends are bumped by different threads
  size_t q0 = 0, q1 = 0;
  std::vector<uint32_t> starts, slice, query;
};
struct ALeadKey {  // sort key of one lead of the shared-intersection group
  uint64_t k1;    // leader handle << 8 | cache
  uint64_t mask;  // doc-matrix bits of the other lists
  uint64_t sig;   // hash of the whole query (lists, weights, k): identical queries become neighbours
  uint32_t q, pad;
};
struct ShareKey {  // one (query, list) pair of the shared-union group
  uint64_t key;    // list position i << 56 | blocks of the term (rare terms first) << 32 | cache
  uint32_t term, q;
};
// launch groups of a batch: 0 AND over bitmap lists, 1 unions, 2 phrases, 3 AND over any lists, 4 boolean
// queries, 5 shared unions, 6 phrase sweep, 7 doc-major unions, 8 shared intersections
constexpr int kNGroups = 9;
struct QuerySlab {  // one slab of a batch's queries, planned by one thread into groups of its own
  Group groups[kNGroups];
  uint32_t n_thr_rows = 0;
  uint64_t algo_bytes = 0;
  bool phrase_all_dense = true;
  int rc = 0;
  std::string err;
};
struct PlanScratch {
  Group groups[kNGroups];
  std::vector<uint32_t> q_cache;      // per query of the batch: its Bm25Weight cache
  std::vector<QuerySlab> q_slabs;
  // doc-major union group (tq_xunion.hip): the lists of the batch (<-> rows of the tile), the queries
  std::vector<TqkDenseRow> xrows;
  std::vector<TqkDenseQuery> xqueries;
  std::vector<uint64_t> xrow_term;           // row -> term handle << 32 | weight bits, in order of first use
  std::unordered_map<uint64_t, uint32_t> xrow_of;  // ... -> row
  uint32_t xgrid = 0, x_bitmap_rows = 0, x_tiles_per_task = 1, x_list_stride = 0, x_max_terms = 1;
  // shared-union group (tq_ushare.hip): leads grouped by term, tasks in launch order
  std::vector<ShareKey> share_keys, share_keys2;
  std::vector<uint64_t> sort_keys, sort_keys2;
  std::vector<uint32_t> term_rank, term_distinct;
  std::vector<TqdLead> leads;
  std::vector<uint4> tasks;
  std::vector<uint32_t> share_pairs;  // per query: (task, lead) pairs = result-list appends at most
  // shared-intersection group (tq_ashare.hip): one lead per query, sorted by (leader, cache, mask)
  std::vector<TqdALead> aleads, aleads_unsorted;
  std::vector<ALeadKey> alead_keys, alead_keys2;
  std::vector<uint32_t> alead_bucket, alead_bucket_at, alead_bucket_starts;
  std::vector<uint8_t> alead_same;
  std::vector<uint4> atasks, atasks_unsorted;
  std::vector<uint32_t> atask_pos, apairs, atask_hist, atask_slab_run;
  struct ARun {  // the leads of one (leader, cache)
    uint32_t r0, r1, term, cache, n_blocks, n_groups, per_group, bpt, nb_warm, n_runs;
    size_t task0;
  };
  std::vector<ARun> aruns;
  uint32_t a_warm_tasks = 0;  // tasks [0, a_warm_tasks) are the warm-up launch
  std::vector<uint32_t> q_leader;        // per query of the batch: the list that would lead it there, or 0xFFFFFFFF
  std::vector<uint32_t> and_lead_count;  // per term handle: AND queries of the batch it could lead in that launch
  std::vector<uint32_t> term_stamp;      // per term handle: last batch that used the list (unique bytes)
  uint32_t batch_stamp = 0;
  uint32_t share_phase_first[TQD_US_MAX_TERMS + 1];  // tasks of list position i: [first[i], first[i+1])
  uint64_t share_table_base = 0;  // TqdLead::dense_off / tf8_off are relative to this device address
  std::vector<uint32_t> lead_cost, sort_start;
  std::vector<PlanSlab> slabs;
  std::vector<std::pair<uint64_t, uint32_t>> keyed;
  PodVec<TqdQuery> q_tmp;
  std::vector<uint32_t> o_tmp, c_tmp, hist;
  std::vector<uint4> sorted_recs;
};
void tq_free_plan_scratch(PlanScratch *p) { delete p; }

namespace {

// Planner threads (TQ_PLAN_THREADS, default 4, 1 = off): the chunk tables of a large batch are
// built in slabs of queries / slices / records.  The helpers are a process-wide pool of detached
// threads that sleep on a condition variable between jobs (created on first use, never torn
// down: a batch plans in four parallel steps, and spawning threads for each of them cost more
// than the steps themselves — 2.1 ms of host time per 10 000-query AND batch against 1.3 ms for
// the same tables built by one thread).  One job at a time: a caller that finds the pool busy
// (another segment planning on another thread) runs its slabs itself.
static uint32_t plan_threads();
class PlanPool {
 public:
  static PlanPool &get() {
    static PlanPool *pool = new PlanPool();  // (leaked on purpose: its threads outlive static destruction)
    return *pool;
  }
  // fn(ctx, slab) for slab in [0, n): the caller takes part, returns when all slabs are done
  void run(uint32_t n, void (*fn)(void *, uint32_t), void *ctx) {
    std::unique_lock<std::mutex> job_lock(job_mutex_, std::try_to_lock);
    if (!job_lock.owns_lock() || !ensure_workers(std::min<uint32_t>(n, plan_threads()) - 1u)) {
      for (uint32_t i = 0; i < n; ++i) fn(ctx, i);
      return;
    }
    uint64_t gen;
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = fn;
      ctx_ = ctx;
      n_ = n;
      gen = generation_.load(std::memory_order_relaxed) + 1;
      done_.store(0, std::memory_order_relaxed);
      ticket_.store(gen << 32, std::memory_order_release);
      generation_.store(gen, std::memory_order_release);
    }
    cv_.notify_all();
    work(gen, fn, ctx, n);
    // (the slabs are short: spin for the last ones instead of sleeping)
    while (done_.load(std::memory_order_acquire) < n) std::this_thread::yield();
  }

 private:
  // Slabs are handed out through one word, generation << 32 | next slab: a helper that wakes up
  // late (its job already over, maybe the next one under way) finds another generation there and
  // takes nothing.
  void work(uint64_t gen, void (*fn)(void *, uint32_t), void *ctx, uint32_t n) {
    for (;;) {
      uint64_t cur = ticket_.load(std::memory_order_acquire);
      if ((cur >> 32) != (gen & 0xFFFFFFFFull) || (uint32_t)cur >= n) return;
      if (!ticket_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
      fn(ctx, (uint32_t)cur);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  bool ensure_workers(uint32_t want) {  // (under job_mutex_)
    while (n_workers_ < want) {
      try {
        std::thread([this] { worker(); }).detach();
        ++n_workers_;
      } catch (...) {  // a thread limit: plan with what there is
        break;
      }
    }
    return n_workers_ > 0;
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      void (*fn)(void *, uint32_t);
      void *ctx;
      uint32_t n;
      // a batch brings a dozen jobs within a millisecond: stay awake for a while after each one (a
      // wake-up through the condition variable costs 50-100 us, more than most of the jobs)
      const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(400);
      while (generation_.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < spin_until)
        __builtin_ia32_pause();
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_.load(std::memory_order_relaxed) != seen; });
        seen = generation_.load(std::memory_order_relaxed);
        fn = fn_;
        ctx = ctx_;
        n = n_;
      }
      work(seen, fn, ctx, n);
    }
  }
  std::mutex job_mutex_, m_;
  std::condition_variable cv_;
  void (*fn_)(void *, uint32_t) = nullptr;
  void *ctx_ = nullptr;
  uint32_t n_ = 0, n_workers_ = 0;
  std::atomic<uint64_t> generation_{0};
  std::atomic<uint64_t> ticket_{0};
  std::atomic<uint32_t> done_{0};
};
template <typename F>
static void parallel_slabs(uint32_t n_slabs, F &&fn) {  // fn(slab) for slab in [0, n_slabs)
  if (n_slabs <= 1) {
    if (n_slabs) fn(0u);
    return;
  }
  PlanPool::get().run(
      n_slabs, [](void *c, uint32_t i) { (*static_cast<typename std::remove_reference<F>::type *>(c))(i); }, (void *)&fn);
}

static uint32_t tune_u32(const char *name, uint32_t dflt) {
  const char *v = getenv(name);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}
// doc-range slices of the launch order / chunks per AND launch (TQ_SLICES, TQ_CHUNKS: tuning only)
static const uint32_t kSlices = std::min<uint32_t>(256u, std::max<uint32_t>(1u, tune_u32("TQ_SLICES", 128)));
// candidate-driven OR cost model: lists whose suffix weight is below kOrDeadFrac of the total are
// expected to be skipped at run time and weigh 1/kOrDeadDiv of a live tile
// (measured on the or5 / mixed batches, kernel ms: 75 % 4.45 / 16.0, 60 % 4.11 / 14.1, 50 % 4.17 / 13.9,
// 40 % 3.92 / 13.8, 30 % 4.03 / 14.9; divisor 4 and 16 both worse than 8)
static const float kOrDeadFrac = (float)tune_u32("TQ_OR_DEAD_PCT", 40) / 100.0f;
static const uint32_t kOrDeadDiv = std::max<uint32_t>(1u, tune_u32("TQ_OR_DEAD_DIV", 8));
static const uint32_t kAndChunks = std::max<uint32_t>(256u, tune_u32("TQ_CHUNKS", 131072));
// candidate unions: chunks per launch as a multiple of kAndChunks (k > 16 / k <= 16)
// candidate unions: doc-range sub-slices per leader list in the launch order
static const uint32_t kOrSubSlices = std::min<uint32_t>(256u, std::max<uint32_t>(1u, tune_u32("TQ_OR_SUBSLICES", 64)));
static const bool kOrSubMajor = tune_u32("TQ_OR_SUBMAJOR", 0) != 0;
static const bool kOrSortQueries = tune_u32("TQ_OR_SORT", 1) != 0;
static uint32_t plan_threads() {
  // (default 1 since round 4: with the per-query sorts gone and the intersections planned per leader the
  // calling thread plans a 10 000-query batch in about a millisecond; helper threads were no faster on
  // any bench workload and a descheduled helper — the GPU box shares its cores — stalled a batch for
  // up to 70 ms)
  static const uint32_t n = std::min<uint32_t>(16u, std::max<uint32_t>(1u, tune_u32("TQ_PLAN_THREADS", 1)));
  return n;
}
// batches below this many chunks are planned by the calling thread alone (TQ_PLAN_PAR_MIN: tests)
static const uint32_t kPlanParMin = tune_u32("TQ_PLAN_PAR_MIN", 16384);
static const uint32_t kOrChunkMul = std::max<uint32_t>(1u, tune_u32("TQ_OR_CHUNK_MUL", 4));
static const uint32_t kOrChunkMulSmallK = std::max<uint32_t>(1u, tune_u32("TQ_OR_CHUNK_MUL_SMALLK", 8));
// boolean queries (the union kernel's BOOL instantiation): measured on the bench shapes, kernel / host ms
// per 2000 queries: x8 6.05 / 4.26, x4 6.02 / 3.48, x2 6.04 / 1.77, x1 6.32 / 1.13 — the chunk records
// and partial lists of 1 M chunks bought nothing
static const uint32_t kBoolChunkMul = std::max<uint32_t>(1u, tune_u32("TQ_BOOL_CHUNK_MUL", 2));

// stable sort of a handful of items (<= TQ_MAX_TERMS): std::stable_sort allocates a buffer per call,
// which was a quarter of the per-query planning time of a 10 000-query batch
template <typename T, typename Less>
inline void small_stable_sort(T *first, T *last, Less less) {
  for (T *i = first + (first != last); i < last; ++i) {
    T v = *i;
    T *j = i;
    while (j > first && less(v, j[-1])) {
      *j = j[-1];
      --j;
    }
    *j = v;
  }
}

int kpl_for(uint32_t k) { return k <= 64 ? 1 : (k <= 128 ? 2 : (k <= 256 ? 4 : 16)); }

// tiles -> chunks of one launch group: runs of consecutive tiles of about equal estimated cost,
// their launch order (doc-range slices) and the number of partial lists per query
int build_group_chunks(Group &g, bool or_windows, PlanScratch &ps, bool boolean_group = false) {
  static const bool ptrace = getenv("TQ_PLAN_TRACE") != nullptr;  // phase times of the planner
  auto pt_last = std::chrono::steady_clock::now();
  auto pt = [&](const char *what) {
    if (!ptrace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[tq plan] %-12s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - pt_last).count());
    pt_last = now;
  };
  g.kpl = kpl_for(g.max_k);
  // Candidate unions: queries that lead with the same lists gather the same doc-matrix rows and
  // decode the same blocks.  Inside a (leader, doc sub-slice) bucket of the launch order the chunks
  // follow the query order, so the queries are put in the order of their leading terms: chunks
  // of one term run next to each other in time and find each other's lines in the L2.  (Results
  // go to their rows through out_index; the order of a group's queries is nobody's business.)
  if (g.mode == TQ_MODE_OR && !or_windows && kOrSortQueries && g.queries.size() > 1) {
    const size_t n = g.queries.size();
    std::vector<std::pair<uint64_t, uint32_t>> &keyed = ps.keyed;  // (leading terms, query)
    keyed.resize(n);
    for (size_t i = 0; i < n; ++i) {
      const TqdQuery &q = g.queries[i];
      keyed[i] = {((uint64_t)q.term[0] << 40) | ((uint64_t)(q.n_terms > 1 ? q.term[1] & 0xFFFFFu : 0u) << 20) |
                      (uint64_t)(q.n_terms > 2 ? q.term[2] & 0xFFFFFu : 0u),
                  (uint32_t)i};
    }
    std::sort(keyed.begin(), keyed.end());  // (ties fall back to the query index: stable)
    PodVec<TqdQuery> &q2 = ps.q_tmp;
    std::vector<uint32_t> &o2 = ps.o_tmp, &c2 = ps.c_tmp;
    q2.resize(n);
    o2.resize(n);
    c2.resize(n);
    for (size_t i = 0; i < n; ++i) {
      q2[i] = g.queries[keyed[i].second];
      o2[i] = g.out_index[keyed[i].second];
      c2[i] = g.tile_cost[keyed[i].second];
    }
    g.queries.swap(q2);
    g.out_index.swap(o2);
    g.tile_cost.swap(c2);
  }
  pt("sort queries");
  g.tile_starts.resize(g.queries.size() + 1);
  uint64_t acc = 0;
  for (size_t i = 0; i < g.queries.size(); ++i) {
    g.tile_starts[i] = (uint32_t)acc;
    g.queries[i].tile_start = (uint32_t)acc;
    acc += g.queries[i].n_tiles;
    if (acc > 0x7FFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (tiles)");
  }
  g.tile_starts[g.queries.size()] = (uint32_t)acc;
  g.total_tiles = (uint32_t)acc;
  // chunks = runs of consecutive tiles of about equal estimated cost; one chunk is one
  // wavefront (AND, phrase) or one workgroup (OR) and the hardware dispatcher hands them out
  // as slots free up, so many small chunks balance the load
  // doc-range slices of the launch order (phrase batches with 64-block tiles: 64 slices, 1.85 ms;
  // 32: 1.90, 128: 1.86, 8: 2.20)
  static const uint32_t kPhSlices = std::min<uint32_t>(256u, std::max<uint32_t>(1u, tune_u32("TQ_PH_SLICES", 64)));
  const uint32_t n_slices = g.mode == TQ_MODE_PHRASE ? std::min<uint32_t>(kSlices, kPhSlices) : kSlices;
  const bool or_win = g.mode == TQ_MODE_OR && or_windows;
  const bool or_cand = g.mode == TQ_MODE_OR && !or_windows;
  // candidate-driven OR: the tiles of a list that MaxScore will most likely find non-essential
  // (the weights of lists i.. together below ~40 % of the query's total weight: top-k docs hold
  // most of the terms) are skipped whole at run time => weigh them as 1/8 of a live tile, so
  // that chunks are sized by the work that is really done.  A query's tiles form runs of equal
  // cost: one per leader (candidate unions) or one for the whole query; every loop below walks
  // runs, never single tiles.
  std::vector<uint32_t> &lead_cost = ps.lead_cost;  // [query][TQ_MAX_TERMS]
  if (or_cand) {
    lead_cost.resize(g.queries.size() * TQ_MAX_TERMS);
    for (size_t qi = 0; qi < g.queries.size(); ++qi) {
      const TqdQuery &dq = g.queries[qi];
      const uint32_t tc = std::max<uint32_t>(1u, g.tile_cost[qi]);
      const bool pruning = (dq.flags & TQD_QF_PRUNE) != 0u;
      float total = 0.0f;
      for (uint32_t m = 0; m < dq.n_terms; ++m) total += dq.weight[m];
      float suffix = total;
      for (uint32_t li = 0; li < dq.n_terms; ++li) {
        lead_cost[qi * TQ_MAX_TERMS + li] =
            (pruning && suffix < kOrDeadFrac * total) ? std::max<uint32_t>(1u, tc / kOrDeadDiv) : tc;
        suffix -= dq.weight[li];
      }
    }
  }
  // run of equal cost that holds tile t of query qi (li = leader of the run, advanced by the
  // caller's cursor: tiles are visited in order)
  auto run_of = [&](size_t qi, uint32_t t, uint32_t &li, uint32_t &run_end) -> uint32_t {
    const TqdQuery &dq = g.queries[qi];
    if (!or_cand) {
      run_end = dq.n_tiles;
      return std::max<uint32_t>(1u, g.tile_cost[qi]);
    }
    while (li + 1u < dq.n_terms && dq.lead_tile_start[li + 1u] <= t) ++li;
    run_end = std::max<uint32_t>(t + 1u, std::min<uint32_t>(dq.n_tiles, dq.lead_tile_start[li + 1u]));
    return lead_cost[qi * TQ_MAX_TERMS + li];
  };
  uint64_t total_cost = 0;
  for (size_t i = 0; i < g.queries.size(); ++i) {
    uint32_t li = 0;
    for (uint32_t t = 0; t < g.queries[i].n_tiles;) {
      uint32_t e;
      const uint32_t tc = run_of(i, t, li, e);
      total_cost += (uint64_t)(e - t) * tc;
      t = e;
    }
  }
  // candidate unions: smaller chunks balance better (the work per tile swings with the
  // threshold); with large k the partial lists (1 KB per chunk and query) and the host's
  // planning time per chunk weigh more
  const uint64_t n_target =
      or_win ? 8192u
             : (or_cand ? (boolean_group ? kBoolChunkMul : (g.max_k <= 16u ? kOrChunkMulSmallK : kOrChunkMul)) * kAndChunks
                        : kAndChunks);
  const uint64_t cost_target = std::max<uint64_t>(or_win ? 1u : 128u,
                                                  (total_cost + n_target - 1) / n_target);
  pt("costs");
  const uint32_t per_chunk = or_win ? TQD_WAVES_PER_WG : 1u;
  const uint32_t li_cap = n_slices * 8u / kOrSubSlices - 1u;
  // The queries are cut into slabs of about equal cost; every slab builds its chunks on its own
  // (a chunk never spans two slabs) and the tables are concatenated afterwards.
  const size_t nq = g.queries.size();
  const uint32_t n_slabs = (uint32_t)std::max<size_t>(1, std::min<size_t>(total_cost / cost_target >= kPlanParMin ? plan_threads() : 1u, nq));
  using Slab = PlanSlab;
  std::vector<Slab> &slabs = ps.slabs;
  if (slabs.size() < n_slabs) slabs.resize(n_slabs);
  {
    size_t qi = 0;
    uint64_t acc_cost = 0;
    for (uint32_t sb = 0; sb < n_slabs; ++sb) {
      slabs[sb].q0 = qi;
      const uint64_t upto = total_cost * (sb + 1) / n_slabs;
      while (qi < nq && (acc_cost < upto || sb + 1 == n_slabs)) {
        uint32_t li = 0;
        for (uint32_t t = 0; t < g.queries[qi].n_tiles;) {
          uint32_t e;
          const uint32_t tc = run_of(qi, t, li, e);
          acc_cost += (uint64_t)(e - t) * tc;
          t = e;
        }
        ++qi;
      }
      slabs[sb].q1 = qi;
    }
    slabs[n_slabs - 1].q1 = nq;
  }
  pt("slabs");
  parallel_slabs(n_slabs, [&](uint32_t sb) {
    Slab &S = slabs[sb];
    S.starts.clear();
    S.slice.clear();
    S.query.clear();
    uint64_t cur_cost = 0;
    bool open_chunk = false;
    uint32_t tc_seen = 0, per_fresh = 1;
    for (size_t i = S.q0; i < S.q1; ++i) {
      TqdQuery &dq = g.queries[i];
      dq.part_start = 0;
      dq.n_parts = 0;
      dq.chunk_first = 0;
      if (!dq.n_tiles) continue;
      uint32_t first_chunk = 0xFFFFFFFFu;
      uint32_t li = 0, li_seen = 0xFFFFFFFFu;
      double sub_scale = 0.0;  // sub-slices per tile of the current leader's run
      const double slice_scale = (double)(n_slices * 8u) / (double)dq.n_tiles;
      for (uint32_t t = 0; t < dq.n_tiles;) {
        uint32_t run_end;
        const uint32_t tc = run_of(i, t, li, run_end);
        if (!open_chunk || cur_cost >= cost_target) {
          S.starts.push_back(dq.tile_start + t);
          S.query.push_back((uint32_t)i);
          // which part of the doc-id space the chunk starts in (lists are spread over it)
          if (or_cand) {
            // candidate-driven OR: high-weight lists first (their matches raise the threshold
            // that lets the tiles of the dense low-weight lists be skipped), doc order inside
            if (li != li_seen) {
              li_seen = li;
              const uint32_t span = std::max<uint32_t>(1u, dq.lead_tile_start[li + 1u] - dq.lead_tile_start[li]);
              sub_scale = (double)kOrSubSlices / (double)span;
            }
            const uint32_t sub = std::min<uint32_t>(kOrSubSlices - 1u, (uint32_t)((double)(t - dq.lead_tile_start[li]) * sub_scale));
            const uint32_t lic = std::min<uint32_t>(li, li_cap);
            S.slice.push_back(std::min<uint32_t>(
                n_slices * 8u - 1u, kOrSubMajor ? sub * (li_cap + 1u) + lic : lic * kOrSubSlices + sub));
          } else {
            S.slice.push_back(std::min<uint32_t>(n_slices * 8u - 1u, (uint32_t)((double)t * slice_scale)));
          }
          cur_cost = 0;
          open_chunk = true;
        }
        if (first_chunk == 0xFFFFFFFFu) first_chunk = (uint32_t)S.starts.size() - 1u;
        // as many tiles of this query as the chunk still takes (a fresh chunk takes the same
        // number of tiles all along a run: the division is per run, not per chunk)
        if (tc != tc_seen) {
          tc_seen = tc;
          per_fresh = (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (cost_target + tc - 1) / tc);
        }
        const uint64_t room = cost_target - cur_cost;
        uint32_t take = cur_cost == 0 ? per_fresh
                                      : (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, (room + tc - 1) / tc);
        take = std::min<uint32_t>(take, run_end - t);
        take = std::max<uint32_t>(take, 1u);
        cur_cost += (uint64_t)take * tc;
        t += take;
      }
      dq.chunk_first = first_chunk;  // slab-local: rebased below
      dq.n_parts = ((uint32_t)S.starts.size() - first_chunk) * per_chunk;
    }
  });
  pt("chunk loop");
  // The slabs' tables stay where they are: chunk c = offs[slab] + index in the slab.  Records
  // {first tile, end tile, first query, chunk} are built slab by slab (sequential reads) and
  // scattered straight to their launch positions' buckets.
  std::vector<size_t> offs(n_slabs + 1, 0);
  for (uint32_t sb = 0; sb < n_slabs; ++sb) offs[sb + 1] = offs[sb] + slabs[sb].starts.size();
  g.n_chunks = (uint32_t)offs[n_slabs];
  for (uint32_t sb = 1; sb < n_slabs; ++sb)
    if (offs[sb])
      for (size_t i = slabs[sb].q0; i < slabs[sb].q1; ++i)
        if (g.queries[i].n_tiles) g.queries[i].chunk_first += (uint32_t)offs[sb];
  auto end_tile_after = [&](uint32_t sb) -> uint32_t {  // first tile of the next non-empty slab
    for (uint32_t nx = sb + 1; nx < n_slabs; ++nx)
      if (!slabs[nx].starts.empty()) return slabs[nx].starts[0];
    return g.total_tiles;
  };
  auto record_of = [&](const Slab &S, uint32_t sb, size_t j, uint32_t slab_end) {
    return make_uint4(S.starts[j], j + 1 < S.starts.size() ? S.starts[j + 1] : slab_end, S.query[j],
                      (uint32_t)(offs[sb] + j));
  };
  g.chunk_recs.resize(g.n_chunks);
  if (or_win) {  // the window kernel runs in chunk order
    for (uint32_t sb = 0; sb < n_slabs; ++sb) {
      const uint32_t slab_end = end_tile_after(sb);
      for (size_t j = 0; j < slabs[sb].starts.size(); ++j)
        g.chunk_recs[offs[sb] + j] = record_of(slabs[sb], sb, j, slab_end);
    }
    pt("records");
    return TQ_OK;
  }
  // Launch order: all chunks of doc-range slice 0 (of every query), then slice 1, ...  The
  // dispatcher hands out workgroups in index order, so at any moment the whole chip works on
  // the same ~1/128 of the doc-id space: the fieldnorm bytes, bitmap words and hot posting
  // blocks of that slice stay in the 4 MB L2s across queries instead of being re-fetched.
  // Inside a slice the chunks are dealt round-robin from its 8 sub-slices: workgroup i runs
  // on XCD i % 8 (observed placement, MI355X_MICROARCH.md), so each XCD's L2 sees one eighth
  // of the slice.  Placement is a speed-up only; nothing depends on it.
  {
    const uint32_t nb = n_slices * 8u;
    std::vector<uint32_t> &start = ps.sort_start;
    // stable counting sort by slice: every slab counts its own histogram, the (slice, slab)
    // prefix sums give every slab its own output positions
    std::vector<uint32_t> &hist = ps.hist;  // [slab][nb]
    hist.assign((size_t)n_slabs * nb, 0);
    parallel_slabs(n_slabs, [&](uint32_t sb) {
      uint32_t *h = hist.data() + (size_t)sb * nb;
      for (uint32_t sl : slabs[sb].slice) ++h[sl];
    });
    start.assign(nb + 1, 0);
    {
      uint32_t run = 0;
      for (uint32_t i = 0; i < nb; ++i) {
        start[i] = run;
        for (uint32_t sb = 0; sb < n_slabs; ++sb) {
          const uint32_t n = hist[(size_t)sb * nb + i];
          hist[(size_t)sb * nb + i] = run;  // becomes the slab's write position in slice i
          run += n;
        }
      }
      start[nb] = run;
    }
    std::vector<uint4> &sorted_recs = ps.sorted_recs;
    sorted_recs.resize(g.n_chunks);
    parallel_slabs(n_slabs, [&](uint32_t sb) {
      const Slab &S = slabs[sb];
      const uint32_t slab_end = end_tile_after(sb);
      uint32_t *h = hist.data() + (size_t)sb * nb;
      for (size_t j = 0; j < S.starts.size(); ++j) sorted_recs[h[S.slice[j]]++] = record_of(S, sb, j, slab_end);
    });
    pt("count sort");
    // slice sl writes chunk_recs[start[8 sl] .. start[8 sl + 8)): slices are independent
    const uint32_t deal_slabs = g.n_chunks >= kPlanParMin ? std::min<uint32_t>(plan_threads(), n_slices) : 1u;
    parallel_slabs(deal_slabs, [&](uint32_t sb) {
      const uint32_t sl0 = (uint32_t)((uint64_t)n_slices * sb / deal_slabs);
      const uint32_t sl1 = (uint32_t)((uint64_t)n_slices * (sb + 1) / deal_slabs);
      for (uint32_t sl = sl0; sl < sl1; ++sl) {
        uint32_t out = start[sl * 8];
        uint32_t at[8], end[8], left = 0;
        for (uint32_t x = 0; x < 8; ++x) {
          at[x] = start[sl * 8 + x];
          end[x] = start[sl * 8 + x + 1];
          left += end[x] - at[x];
        }
        while (left) {
          for (uint32_t x = 0; x < 8; ++x) {
            if (at[x] < end[x]) {
              g.chunk_recs[out++] = sorted_recs[at[x]++];
              --left;
            } else if (left) {  // keep the i % 8 alignment: borrow from the fullest sub-slice
              uint32_t best = 8, most = 0;
              for (uint32_t y = 0; y < 8; ++y)
                if (end[y] - at[y] > most) {
                  most = end[y] - at[y];
                  best = y;
                }
              if (best < 8) {
                g.chunk_recs[out++] = sorted_recs[--end[best]];
                --left;
              }
            }
          }
        }
      }
    });
  }
  pt("deal");
  pt("records");
  return TQ_OK;
}

// The shared-union launch (tq_ushare.hip): the (query, list) pairs of the group's pure unions are
// sorted by term, cut into groups of <= TQD_US_GROUP leads, and every group gets one task per run of
// blocks of its term.  Rare (high-weight) terms come first in the task order: their matches raise
// the thresholds that let the tasks of the dense terms end at their first look at them.
int build_share_plan(tq_segment *s, Group &g, PlanScratch &ps) {
  static const uint32_t kTaskCost = std::max<uint32_t>(64u, tune_u32("TQ_US_TASK_COST", 2048));
  static const uint32_t kTaskBlocksMax = std::max<uint32_t>(1u, tune_u32("TQ_US_TASK_BLOCKS", 64));
  static const uint32_t kGroupMax = std::min<uint32_t>(TQD_US_GROUP, std::max<uint32_t>(1u, tune_u32("TQ_US_GROUP", TQD_US_GROUP)));
  const size_t nq = g.queries.size();
  g.kpl = kpl_for(g.max_k);
  static const bool ptrace = getenv("TQ_PLAN_TRACE") != nullptr;  // phase times of the planner
  auto pt_last = std::chrono::steady_clock::now();
  auto pt = [&](const char *what) {
    if (!ptrace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[tq share plan] %-12s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - pt_last).count());
    pt_last = now;
  };
  // Leads by list position first: position 0 is every query's highest-weight list, and its docs
  // settle the query's threshold — all tasks of position i are launched (and done) before those of
  // position i + 1 (one launch per position).  Inside a position: by term, rare terms first.
  // The order is (position, blocks of the term, term, cache, query).  The batch's distinct terms are
  // ranked by (blocks, handle) first — a few hundred — and the pairs, generated in query order, go
  // through a stable radix sort on position | rank | cache (a comparison sort of the 25 000 pairs of
  // a 5000-query batch was two thirds of this function's time).
  std::vector<ShareKey> &keys = ps.share_keys;
  {
    std::vector<uint32_t> &rank = ps.term_rank, &distinct = ps.term_distinct;
    if (rank.size() < s->terms.size()) rank.resize(s->terms.size(), 0u);
    distinct.clear();
    size_t n_pairs = 0;
    for (size_t q = 0; q < nq; ++q) {
      const TqdQuery &dq = g.queries[q];
      n_pairs += dq.n_terms;
      for (uint32_t i = 0; i < dq.n_terms; ++i)
        if (rank[dq.term[i]] != 0xFFFFFFFFu) {  // (0xFFFFFFFF = seen in this batch; reset below)
          rank[dq.term[i]] = 0xFFFFFFFFu;
          distinct.push_back(dq.term[i]);
        }
    }
    std::sort(distinct.begin(), distinct.end(), [&](uint32_t a, uint32_t b) {
      const uint32_t na = s->terms[a].n_blocks, nb = s->terms[b].n_blocks;
      return na != nb ? na < nb : a < b;
    });
    for (size_t r = 0; r < distinct.size(); ++r) rank[distinct[r]] = (uint32_t)r;
    std::vector<uint64_t> &sk = ps.sort_keys, &sk2 = ps.sort_keys2;
    std::vector<ShareKey> &k2 = ps.share_keys2;
    keys.resize(n_pairs);
    k2.resize(n_pairs);
    sk.resize(n_pairs);
    sk2.resize(n_pairs);
    size_t at = 0;
    for (size_t q = 0; q < nq; ++q) {
      const TqdQuery &dq = g.queries[q];
      for (uint32_t i = 0; i < dq.n_terms; ++i, ++at) {
        const uint64_t nb = std::min<uint64_t>(0xFFFFFFu, s->terms[dq.term[i]].n_blocks);
        keys[at] = {((uint64_t)i << 56) | (nb << 32) | (uint64_t)(dq.cache_idx & 0xFFu), dq.term[i], (uint32_t)q};
        sk[at] = ((uint64_t)i << 40) | ((uint64_t)rank[dq.term[i]] << 8) | (uint64_t)(dq.cache_idx & 0xFFu);
      }
    }
    for (uint32_t t : distinct) rank[t] = 0u;  // (any value but the marker)
    for (uint32_t shift = 0; shift < 48; shift += 8) {  // LSD, one byte per pass; stable: queries stay in order
      uint32_t hist[257] = {0};
      for (size_t i = 0; i < n_pairs; ++i) ++hist[((sk[i] >> shift) & 0xFFu) + 1u];
      bool one_bucket = false;
      for (uint32_t d = 0; d < 256; ++d) one_bucket = one_bucket || hist[d + 1] == n_pairs;
      if (one_bucket) continue;  // every key has the same byte here
      for (uint32_t d = 0; d < 256; ++d) hist[d + 1] += hist[d];
      for (size_t i = 0; i < n_pairs; ++i) {
        const uint32_t o = hist[(sk[i] >> shift) & 0xFFu]++;
        sk2[o] = sk[i];
        k2[o] = keys[i];
      }
      sk.swap(sk2);
      keys.swap(k2);
    }
  }
  pt("keys + sort");
  std::vector<TqdLead> &leads = ps.leads;
  std::vector<uint4> &tasks = ps.tasks;
  std::vector<uint32_t> &pairs = ps.share_pairs;
  leads.resize(keys.size());
  tasks.clear();
  pairs.assign(nq, 0u);
  // bitmaps and byte-wide tfs are addressed as 32-bit offsets (8-byte units) from one base: the
  // lowest table address of the segment (the caller checked that they span less than 32 GB:
  // otherwise the unions keep the per-query kernel)
  ps.share_table_base = s->share_table_lo;
  auto off_of = [&](const void *ptr) -> uint32_t {
    return ptr ? (uint32_t)(((uint64_t)ptr - ps.share_table_base) >> 3) : 0u;
  };
  auto column_of = [&](uint32_t handle) -> uint32_t {  // doc-matrix bit of the list, or 0
    const uint32_t slot1 = (s->h_dterms[handle].has_freq >> 8) & 0xFFu;
    return slot1 ? 8u + (slot1 - 1u) : 0u;
  };
  // (a lead reads only its own query: the table is filled by the planner's threads, a slab each)
  const uint32_t lead_slabs = keys.size() >= 8192 ? plan_threads() : 1u;
  parallel_slabs(lead_slabs, [&](uint32_t sb) {
  const size_t at0 = keys.size() * sb / lead_slabs, at1 = keys.size() * (sb + 1) / lead_slabs;
  for (size_t at = at0; at < at1; ++at) {
    const ShareKey &k = keys[at];
    const uint32_t li = (uint32_t)(k.key >> 56);
    const TqdQuery &dq = g.queries[k.q];
    TqdLead ld{};
    ld.query = k.q;
    ld.w = dq.weight[li];
    uint32_t ncols = 0, nocol = 0, nopc = 0, uses_sig = 0;
    float suffix = 0.0f, sparse_after = 0.0f;
    for (uint32_t m = dq.n_terms; m-- > li;) suffix += dq.weight[m];
    for (uint32_t m = 0; m < dq.n_terms; ++m) {
      const uint32_t col = column_of(dq.term[m]);
      // lists without a column: their signature bit (docsig), if the segment keeps signatures
      const uint32_t sig1 = !col ? (s->h_dterms[dq.term[m]].has_freq >> 16) & 0xFFu : 0u;
      ld.sig[m] = (uint8_t)sig1;
      if (!col) nocol |= 1u << m;
      if (!col && !sig1) nopc |= 1u << m;
      if (sig1 && m != li) uses_sig = 1;
      if (m < li) {
        if (col) ld.before_mask |= 1ull << col;
      } else if (m > li) {
        ld.dense_off[m - li - 1u] = off_of(s->opt.use_dense ? s->terms[dq.term[m]].dense_blob : nullptr);
        ld.tf8_off[m - li - 1u] = off_of(s->terms[dq.term[m]].tf8_blob);
        const uint32_t bitpos = col ? col : (sig1 ? TQD_SIG_SHIFT + (sig1 - 1u) : 0u);
        if (bitpos) {
          if (ncols < 4u)
            ld.cols_lo |= bitpos << (8u * ncols);
          else
            ld.cols_hi |= bitpos << (8u * (ncols - 4u));
          ld.aw[ncols] = dq.weight[m];
          ++ncols;
        } else {
          sparse_after += dq.weight[m];
        }
      }
    }
    ld.suffix = suffix;
    ld.sparse_after = sparse_after;
    ld.info = li | (ncols << 4) | (dq.n_terms << 8) | (uses_sig << 12) | (nocol << 16) | (nopc << 24);
    leads[at] = ld;
  }
  });
  pt("leads");
  // cost of every position's tasks together (a block costs its decode + one test per lead): a
  // position with little work is cut into smaller tasks, so that it still fills the chip and its
  // launch does not end on a few long tasks
  // (measured, kernel ms: or5 at k = 100 wants ~6144 tasks per position — 3072: 2.65, 4096: 2.54, 5120:
  // 2.44, 6144: 2.36, 7168: 2.51, 10240: 2.83 — the mixed stream at k = 10 ~4096: 3072: 9.75, 4096: 9.02,
  // 5120: 9.19, 6144: 9.54: its thresholds settle after a few docs, and longer tasks keep the
  // feedback inside one wave)
  static const uint32_t kPhaseTasksEnv = tune_u32("TQ_US_PHASE_TASKS", 0);
  const uint32_t kPhaseTasks = kPhaseTasksEnv ? kPhaseTasksEnv : (g.max_k <= 16u ? 4096u : 6144u);
  uint64_t phase_cost[TQD_US_MAX_TERMS] = {};
  for (size_t r0 = 0; r0 < keys.size();) {
    size_t r1 = r0;
    while (r1 < keys.size() && keys[r1].key == keys[r0].key && keys[r1].term == keys[r0].term) ++r1;
    const uint32_t n_run = (uint32_t)(r1 - r0);
    const uint32_t n_groups = (n_run + kGroupMax - 1) / kGroupMax;
    phase_cost[keys[r0].key >> 56] += (uint64_t)s->terms[keys[r0].term].n_blocks * (4u * n_groups + n_run);
    r0 = r1;
  }
  // runs of one (position, term, cache): groups of leads x runs of blocks
  for (uint32_t i = 0; i <= TQD_US_MAX_TERMS; ++i) ps.share_phase_first[i] = 0;
  uint32_t phase = 0;
  for (size_t r0 = 0; r0 < keys.size();) {
    size_t r1 = r0;
    while (r1 < keys.size() && keys[r1].key == keys[r0].key && keys[r1].term == keys[r0].term) ++r1;
    const uint32_t li = (uint32_t)(keys[r0].key >> 56);
    const uint32_t task_cost = (uint32_t)std::min<uint64_t>(kTaskCost, std::max<uint64_t>(36u, phase_cost[li] / kPhaseTasks));
    while (phase < li) ps.share_phase_first[++phase] = (uint32_t)tasks.size();
    const uint32_t term = keys[r0].term, cache = (uint32_t)keys[r0].key & 0xFFu;
    const uint32_t n_blocks = s->terms[term].n_blocks;
    const uint32_t n_run = (uint32_t)(r1 - r0);
    const uint32_t n_groups = (n_run + kGroupMax - 1) / kGroupMax;
    const uint32_t per_group = (n_run + n_groups - 1) / n_groups;
    // blocks per task: about equal cost (a block costs its decode + one test per lead); small
    // tasks keep the share of the batch that is in flight before thresholds exist small
    uint32_t bpt = task_cost / (4u + per_group);
    bpt = std::min<uint32_t>(kTaskBlocksMax, std::max<uint32_t>(1u, bpt));
    for (uint32_t j0 = 0; j0 < n_blocks; j0 += bpt) {
      const uint32_t nb = std::min<uint32_t>(bpt, n_blocks - j0);
      for (uint32_t gr = 0; gr < n_groups; ++gr) {
        const uint32_t l0 = gr * per_group, l1 = std::min<uint32_t>(n_run, l0 + per_group);
        if (l0 >= l1) continue;
        tasks.push_back(make_uint4(term, j0, nb | ((l1 - l0) << 16) | (cache << 24), (uint32_t)r0 + l0));
      }
    }
    const uint32_t n_runs = (n_blocks + bpt - 1) / bpt;
    for (size_t a = r0; a < r1; ++a) pairs[keys[a].q] += n_runs;
    r0 = r1;
  }
  while (phase < TQD_US_MAX_TERMS) ps.share_phase_first[++phase] = (uint32_t)tasks.size();
  pt("tasks");
  if (tasks.size() > 0x7FFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (tasks)");
  // result lists: every (task, lead) pair appends at most k entries
  uint64_t entries = 0;
  for (size_t q = 0; q < nq; ++q) {
    TqdQuery &dq = g.queries[q];
    const uint64_t cap = (uint64_t)pairs[q] * dq.k;
    if (entries + cap > 0xFFFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (result lists)");
    dq.part_start = (uint32_t)entries;
    dq.n_parts = (uint32_t)cap;
    dq.chunk_first = 0;
    entries += cap;
  }
  g.total_tiles = (uint32_t)tasks.size();
  g.n_chunks = (uint32_t)tasks.size();
  return TQ_OK;
}

// The shared-intersection launch (tq_ashare.hip): one lead per AND query; the leads of one (leader
// list, Bm25 cache) are sorted by their membership mask (the kernel keeps a block's membership
// ballots across consecutive leads with the same mask), cut into groups of <= TQD_AS_GROUP, and every
// group gets one task per run of blocks of the leader.  Tasks are launched in doc order (all
// leaders' runs of the first 1/4096 of the doc-id space, then the next, ...): the chip works on one
// part of the doc matrix at a time, and every query's threshold rises as its leader is walked.
int build_ashare_plan(tq_segment *s, Group &g, PlanScratch &ps) {
  static const uint32_t kTaskPairsEnv = std::max<uint32_t>(32u, tune_u32("TQ_AS_TASK_PAIRS", 512));
  static const uint32_t kTaskBlocksMax = std::min<uint32_t>(0xFFFFu, std::max<uint32_t>(1u, tune_u32("TQ_AS_TASK_BLOCKS", 64)));
  static const uint32_t kGroupMax = std::min<uint32_t>(TQD_AS_GROUP, std::max<uint32_t>(1u, tune_u32("TQ_AS_GROUP", TQD_AS_GROUP)));
  static const uint64_t kListBudget = (uint64_t)std::max<uint32_t>(1u, tune_u32("TQ_AS_LIST_MB", 1024)) << 20;
  static const bool ptrace = getenv("TQ_PLAN_TRACE") != nullptr;  // phase times of the planner
  auto pt_last = std::chrono::steady_clock::now();
  auto pt = [&](const char *what) {
    if (!ptrace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[tq ashare plan] %-12s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - pt_last).count());
    pt_last = now;
  };
  const size_t nq = g.queries.size();
  g.kpl = g.max_k <= 64 ? 1 : 2;
  auto column_of = [&](uint32_t handle) -> uint32_t {  // doc-matrix bit of the list, or 0
    const uint32_t slot1 = (s->h_dterms[handle].has_freq >> 8) & 0xFFu;
    return slot1 ? 8u + (slot1 - 1u) : 0u;
  };
  ps.share_table_base = s->share_table_lo;
  auto off_of = [&](const void *ptr) -> uint32_t {
    return ptr ? (uint32_t)(((uint64_t)ptr - ps.share_table_base) >> 3) : 0u;
  };
  // ---- leads, in query order (filled by the planner's threads, a slab of queries each), with their
  // sort keys: (leader, cache) | mask | a hash of the whole query (lists, weights, k)
  std::vector<TqdALead> &leads = ps.aleads, &unsorted = ps.aleads_unsorted;
  std::vector<ALeadKey> &keys = ps.alead_keys;
  leads.resize(nq);
  unsorted.resize(nq);
  keys.resize(nq);
  const uint32_t fill_slabs = nq >= 4096 ? plan_threads() : 1u;
  parallel_slabs(fill_slabs, [&](uint32_t sb) {
    const size_t q0 = nq * sb / fill_slabs, q1 = nq * (sb + 1) / fill_slabs;
    for (size_t q = q0; q < q1; ++q) {
      const TqdQuery &dq = g.queries[q];
      TqdALead ld{};
      ld.query = (uint32_t)q;
      ld.w = dq.weight[0];
      float rest = 0.0f;
      uint64_t mask = 0, sig = 0x9E3779B97F4A7C15ull * (uint64_t)(dq.n_terms | (dq.k << 8));
      for (uint32_t m = 0; m < dq.n_terms; ++m) {
        uint32_t wb;
        memcpy(&wb, &dq.weight[m], sizeof wb);
        sig = (sig ^ (((uint64_t)dq.term[m] << 32) | wb)) * 0xFF51AFD7ED558CCDull;
        sig ^= sig >> 29;
        if (!m) continue;
        rest += dq.weight[m];
        const uint32_t col = column_of(dq.term[m]);
        const uint32_t sig1 = !col ? (s->h_dterms[dq.term[m]].has_freq >> 16) & 0xFFu : 0u;
        const uint32_t bitpos = col ? col : (sig1 ? TQD_SIG_SHIFT + (sig1 - 1u) : 0u);
        if (bitpos) mask |= 1ull << bitpos;
      }
      ld.rest = rest;
      ld.mask_lo = (uint32_t)mask;
      ld.mask_hi = (uint32_t)(mask >> 32);
      ld.info = dq.n_terms | (column_of(dq.term[1]) ? 0x100u : 0u);
      ld.dense_off = off_of(s->terms[dq.term[1]].dense_blob);
      ld.tf8_off = off_of(s->terms[dq.term[1]].tf8_blob);
      ld.k = dq.k;
      ld.thr_row = dq.thr_index;
      unsorted[q] = ld;
      keys[q] = ALeadKey{((uint64_t)dq.term[0] << 8) | (uint64_t)(dq.cache_idx & 0xFFu), mask, sig, (uint32_t)q, 0u};
    }
  });
  pt("leads");
  // ---- order: (leader, cache), then mask, then the query hash (identical queries become neighbours:
  // twins), stable in the query index.  Buckets by leader first (a counting sort: a few hundred
  // leaders), then every bucket by the rest of the key — the planner's threads take a share of the
  // buckets each (a comparison sort of the whole table was half of this function's time)
  {
    std::vector<uint32_t> &cnt = ps.alead_bucket;
    const size_t nt = s->terms.size();
    cnt.assign(nt + 1, 0u);
    for (size_t q = 0; q < nq; ++q) ++cnt[(size_t)(keys[q].k1 >> 8) + 1];
    for (size_t t = 0; t < nt; ++t) cnt[t + 1] += cnt[t];
    std::vector<ALeadKey> &tmp = ps.alead_keys2;
    tmp.resize(nq);
    std::vector<uint32_t> &at = ps.alead_bucket_at;
    at.assign(cnt.begin(), cnt.end() - 1);
    for (size_t q = 0; q < nq; ++q) tmp[at[(size_t)(keys[q].k1 >> 8)]++] = keys[q];
    keys.swap(tmp);
    // non-empty buckets, cut into slabs of about equal size
    std::vector<uint32_t> &starts = ps.alead_bucket_starts;
    starts.clear();
    for (size_t t = 0; t < nt; ++t)
      if (cnt[t + 1] > cnt[t]) starts.push_back(cnt[t]);
    starts.push_back((uint32_t)nq);
    const uint32_t n_b = (uint32_t)starts.size() - 1u;
    const uint32_t sort_slabs = nq >= 4096 ? std::min<uint32_t>(plan_threads(), std::max<uint32_t>(1u, n_b)) : 1u;
    parallel_slabs(sort_slabs, [&](uint32_t sb) {
      for (uint32_t b = sb; b < n_b; b += sort_slabs)  // (interleaved: the big buckets are the first leaders)
        std::sort(keys.begin() + starts[b], keys.begin() + starts[b + 1], [](const ALeadKey &a, const ALeadKey &b2) {
          if (a.k1 != b2.k1) return a.k1 < b2.k1;
          if (a.mask != b2.mask) return a.mask < b2.mask;
          if (a.sig != b2.sig) return a.sig < b2.sig;
          return a.q < b2.q;
        });
    });
  }
  pt("sort");
  auto same_query = [&](const ALeadKey &a, const ALeadKey &b) -> bool {  // (the hash only proposes)
    if (a.k1 != b.k1 || a.mask != b.mask || a.sig != b.sig) return false;
    const TqdALead &la = unsorted[a.q], &lb = unsorted[b.q];
    if ((la.info & 31u) != (lb.info & 31u) || la.k != lb.k || memcmp(&la.w, &lb.w, 4) || memcmp(&la.rest, &lb.rest, 4) ||
        la.dense_off != lb.dense_off)
      return false;
    if ((la.info & 31u) == 2u) return true;  // (leader, list 1 — every list has its own bitmap —, both weights, k)
    const TqdQuery &qa = g.queries[a.q], &qb = g.queries[b.q];
    return !memcmp(qa.term, qb.term, qa.n_terms * sizeof(uint32_t)) &&
           !memcmp(qa.weight, qb.weight, qa.n_terms * sizeof(float));
  };
  // identical queries share one row of threshold slots, whatever groups they end up in (a slot is
  // hash(doc): the same doc lands in the same slot whichever group scored it)
  std::vector<uint8_t> &same_as_prev = ps.alead_same;
  same_as_prev.resize(nq);
  const uint32_t gather_slabs = nq >= 4096 ? plan_threads() : 1u;
  parallel_slabs(gather_slabs, [&](uint32_t sb) {
    const size_t i0 = nq * sb / gather_slabs, i1 = nq * (sb + 1) / gather_slabs;
    for (size_t i = i0; i < i1; ++i) {
      leads[i] = unsorted[keys[i].q];
      same_as_prev[i] = i && same_query(keys[i], keys[i - 1]) ? 1 : 0;
    }
  });
  for (size_t i = 1; i < nq; ++i)
    if (same_as_prev[i]) leads[i].thr_row = leads[i - 1].thr_row;
  pt("gather");
  // ---- tasks: groups of leads x runs of blocks; fewer, longer tasks if the result lists (k entries
  // per (task, lead) pair) would not fit the budget.  The first kWarmPermille / 1000 of every leader go
  // out as short tasks in a launch of their own: every resident wavefront starts a launch with the
  // thresholds it finds, and with thresholds of zero the first wavefronts (an eighth of the batch)
  // sent every match through the scoring stage — a warm-up over a fraction of a percent of the blocks
  // leaves the main launch the k-th best of a sample of every query to start from.
  static const uint32_t kWarmPermille = std::min<uint32_t>(1000u, tune_u32("TQ_AS_WARM_PERMILLE", 2));
  static const uint32_t kWarmBlocks = std::max<uint32_t>(1u, tune_u32("TQ_AS_WARM_BLOCKS", 2));
  std::vector<uint4> &tasks = ps.atasks, &raw = ps.atasks_unsorted;
  std::vector<uint32_t> &pos = ps.atask_pos, &pairs = ps.apairs;
  pairs.resize(nq);
  // the runs of one (leader, cache): their groups, task sizes and where their tasks start
  std::vector<PlanScratch::ARun> &runs = ps.aruns;
  uint32_t task_pairs = kTaskPairsEnv;
  size_t n_tasks = 0;
  for (;;) {
    runs.clear();
    n_tasks = 0;
    uint64_t entries = 0;
    for (size_t r0 = 0; r0 < nq;) {
      size_t r1 = r0;
      uint64_t k_sum = 0;
      while (r1 < nq && keys[r1].k1 == keys[r0].k1) k_sum += leads[r1++].k;
      PlanScratch::ARun R;
      R.r0 = (uint32_t)r0;
      R.r1 = (uint32_t)r1;
      R.term = (uint32_t)(keys[r0].k1 >> 8);
      R.cache = (uint32_t)keys[r0].k1 & 0xFFu;
      R.n_blocks = s->terms[R.term].n_blocks;
      const uint32_t n_run = (uint32_t)(r1 - r0);
      const uint32_t n_groups = (n_run + kGroupMax - 1) / kGroupMax;
      R.per_group = (n_run + n_groups - 1) / n_groups;
      R.n_groups = (n_run + R.per_group - 1) / R.per_group;  // (the non-empty ones)
      R.bpt = std::min<uint32_t>(kTaskBlocksMax, std::max<uint32_t>(1u, task_pairs / R.per_group));
      R.nb_warm = (uint32_t)((uint64_t)R.n_blocks * kWarmPermille / 1000u);
      R.n_runs = (R.nb_warm + kWarmBlocks - 1) / kWarmBlocks + (R.n_blocks - R.nb_warm + R.bpt - 1) / R.bpt;
      R.task0 = n_tasks;
      n_tasks += (size_t)R.n_runs * R.n_groups;
      entries += (uint64_t)R.n_runs * k_sum;
      runs.push_back(R);
      r0 = r1;
    }
    if ((entries * sizeof(uint64_t) <= kListBudget && entries <= 0xFFFFFFFFull) || task_pairs >= (1u << 22)) {
      if (entries > 0xFFFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (result lists)");
      break;
    }
    task_pairs *= 2u;
  }
  if (n_tasks > 0x7FFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (tasks)");
  raw.resize(n_tasks);
  pos.resize(n_tasks);
  tasks.resize(n_tasks);
  // slabs of runs of about equal task counts: each fills its tasks and counts them by doc slice; the
  // (slice, slab) prefix sums give every slab its places in the launch order (a stable counting sort:
  // slice 0 = the warm-up launch, then the main launch's tasks by doc slice)
  constexpr uint32_t kSl = 4098;
  const uint32_t t_slabs = n_tasks >= 16384 ? std::min<uint32_t>(plan_threads(), (uint32_t)runs.size()) : 1u;
  std::vector<uint32_t> &hist = ps.atask_hist;
  hist.assign((size_t)t_slabs * kSl, 0u);
  std::vector<uint32_t> &slab_run = ps.atask_slab_run;
  slab_run.assign(t_slabs + 1, (uint32_t)runs.size());
  {
    uint32_t sb = 0;
    slab_run[0] = 0;
    for (uint32_t r = 0; r < runs.size() && sb + 1 < t_slabs; ++r)
      if (runs[r].task0 >= n_tasks * (sb + 1) / t_slabs) slab_run[++sb] = r;
    for (uint32_t x = sb + 1; x < t_slabs; ++x) slab_run[x] = (uint32_t)runs.size();
  }
  parallel_slabs(t_slabs, [&](uint32_t sb) {
    uint32_t *h = hist.data() + (size_t)sb * kSl;
    for (uint32_t r = slab_run[sb]; r < slab_run[sb + 1]; ++r) {
      const PlanScratch::ARun &R = runs[r];
      for (uint32_t a = R.r0; a < R.r1; ++a) {  // twins: the same query as the lead before, inside one group
        const bool twin = (a - R.r0) % R.per_group != 0 && same_as_prev[a];
        leads[a].info = (leads[a].info & ~0x200u) | (twin ? 0x200u : 0u);
        pairs[keys[a].q] = R.n_runs;
      }
      const uint32_t n_run = R.r1 - R.r0;
      const uint64_t slice_mul = ((uint64_t)1 << 44) / R.n_blocks;  // (j0 << 12) / n_blocks without the division
      size_t at = R.task0;
      for (uint32_t j0 = 0; j0 < R.n_blocks;) {
        const bool warm = j0 < R.nb_warm;
        const uint32_t nb = warm ? std::min<uint32_t>(kWarmBlocks, R.nb_warm - j0) : std::min<uint32_t>(R.bpt, R.n_blocks - j0);
        const uint32_t slice = warm ? 0u : 1u + std::min<uint32_t>(4095u, (uint32_t)((j0 * slice_mul) >> 32));
        for (uint32_t gr = 0; gr < R.n_groups; ++gr) {
          const uint32_t l0 = gr * R.per_group, l1 = std::min<uint32_t>(n_run, l0 + R.per_group);
          raw[at] = make_uint4(R.term, j0, nb | ((l1 - l0) << 16) | (R.cache << 24), R.r0 + l0);
          pos[at] = slice;
          ++at;
        }
        h[slice] += R.n_groups;
        j0 += nb;
      }
    }
  });
  {
    uint32_t run = 0;
    for (uint32_t sl = 0; sl < kSl; ++sl)
      for (uint32_t sb = 0; sb < t_slabs; ++sb) {
        const uint32_t n = hist[(size_t)sb * kSl + sl];
        hist[(size_t)sb * kSl + sl] = run;  // becomes the slab's write position in this slice
        run += n;
        if (sl == 0 && sb + 1 == t_slabs) ps.a_warm_tasks = run;
      }
  }
  parallel_slabs(t_slabs, [&](uint32_t sb) {
    uint32_t *h = hist.data() + (size_t)sb * kSl;
    const size_t t0 = slab_run[sb] < runs.size() ? runs[slab_run[sb]].task0 : n_tasks;
    const size_t t1 = slab_run[sb + 1] < runs.size() ? runs[slab_run[sb + 1]].task0 : n_tasks;
    for (size_t i = t0; i < t1; ++i) tasks[h[pos[i]]++] = raw[i];
  });
  pt("tasks");
  uint64_t entries = 0;
  for (size_t q = 0; q < nq; ++q) {
    TqdQuery &dq = g.queries[q];
    const uint64_t cap = (uint64_t)pairs[q] * dq.k;
    if (entries + cap > 0xFFFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (result lists)");
    dq.part_start = (uint32_t)entries;
    dq.n_parts = (uint32_t)cap;
    dq.chunk_first = 0;
    entries += cap;
  }
  g.total_tiles = (uint32_t)tasks.size();
  g.n_chunks = (uint32_t)tasks.size();
  return TQ_OK;
}

// Planning of one TQ_MODE_BOOL query: clause layout of the union kernel (tq_union.hip), pruning
// flags, tile sizes.  An empty result leaves dq.n_terms == 0 and n_tiles == 0.
inline uint64_t xrow_key(uint32_t term, float w) {
  uint32_t wb;
  memcpy(&wb, &w, sizeof wb);
  return ((uint64_t)term << 32) | wb;
}
// The doc-major union group (tq_xunion.hip): rows = the distinct (list, weight) pairs of its queries, those with a
// bitmap first; tasks = runs of 128-doc tiles handed out by an atomic counter to one workgroup per
// CU; every query gets a result list of grid * k entries (a workgroup appends at most k).
int build_dense_plan(tq_segment *s, Group &g, PlanScratch &ps, uint32_t cus) {
  const uint32_t n_rows = (uint32_t)ps.xrow_term.size();
  std::vector<uint32_t> new_row(n_rows, 0u);
  ps.xrows.assign(n_rows, TqkDenseRow{});
  uint32_t n_a = 0;
  for (int pass = 0; pass < 2; ++pass)  // bitmap rows, then the others; first use order inside each
    for (uint32_t r = 0, at = pass ? n_a : 0u; r < n_rows; ++r) {
      const uint32_t h = (uint32_t)(ps.xrow_term[r] >> 32);
      const TermHost &th = s->terms[h];
      const bool bitmap = th.dense_blob && th.tf8_blob;
      if (bitmap != (pass == 0)) continue;
      TqkDenseRow row{};
      row.handle = h;
      row.doc_freq = th.doc_freq;
      {
        const uint32_t wb = (uint32_t)ps.xrow_term[r];
        memcpy(&row.w, &wb, sizeof wb);
      }
      if (bitmap) {
        row.dense = (const uint2 *)th.dense_blob;
        row.tf8 = (const uint8_t *)th.tf8_blob;
        ++n_a;
      } else {
        const size_t doc_bytes = ((size_t)th.doc_freq * sizeof(uint32_t) + 15) & ~(size_t)15;
        row.flat_docs = (const uint32_t *)th.flat_blob;
        row.tf8 = (const uint8_t *)th.flat_blob + doc_bytes;
      }
      new_row[r] = at;
      ps.xrows[at++] = row;
    }
  ps.x_bitmap_rows = n_a;
  const uint32_t n_tiles = (s->max_doc + TQK_XU_TILE - 1) / TQK_XU_TILE;
  static const uint32_t kTaskDiv = std::max<uint32_t>(1u, tune_u32("TQ_XU_TASKS_PER_CU", 16));
  ps.x_tiles_per_task = std::min<uint32_t>(32u, std::max<uint32_t>(1u, n_tiles / (cus * kTaskDiv)));
  const uint32_t n_tasks = (n_tiles + ps.x_tiles_per_task - 1) / ps.x_tiles_per_task;
  ps.xgrid = std::min<uint32_t>(n_tasks, cus);
  g.kpl = g.max_k <= 64 ? 1 : 2;
  g.n_chunks = n_tasks;
  g.total_tiles = n_tiles;
  ps.x_list_stride = ps.xgrid * g.max_k;
  ps.xqueries.assign(g.queries.size(), TqkDenseQuery{});
  ps.x_max_terms = 1;
  for (size_t qi = 0; qi < g.queries.size(); ++qi) {
    TqdQuery &dq = g.queries[qi];
    TqkDenseQuery &xq = ps.xqueries[qi];
    for (uint32_t i = 0; i < 8u; ++i) {  // (beyond n_terms: the all-zero row)
      const uint32_t row = i < dq.n_terms ? new_row[ps.xrow_of[xrow_key(dq.term[i], dq.weight[i])]] : n_rows;
      (i < 4 ? xq.rows_lo : xq.rows_hi) |= row << (8u * (i & 3u));
    }
    ps.x_max_terms = std::max(ps.x_max_terms, dq.n_terms);
    xq.nt_k = dq.n_terms | (dq.k << 8);
    xq.thr_row = dq.thr_index;
    dq.part_start = (uint32_t)(qi * ps.x_list_stride);
    dq.n_parts = ps.x_list_stride;
  }
  return TQ_OK;
}

int plan_bool_query(tq_segment *s, const tq_query &q, uint32_t qi, TqdQuery &dq, uint64_t &qbytes,
                    uint32_t &n_tiles, uint32_t &tile_cost, uint32_t &n_thr_rows, bool exhaustive) {
  // BooleanQuery whose clauses are terms or unions of terms (`+a b -c`, `+a +(b OR c)`),
  // BooleanWeight::complex_scorer (boolean_weight.rs:236-431).  A clause = the terms sharing
  // one clause_of value.  Absent terms are EmptyScorers: they drop out of unions, an empty
  // Must clause empties the query (:249-251), empty Should / MustNot clauses are removed.
  struct Clause {
    uint8_t occur;
    uint32_t id, n = 0, terms[TQ_MAX_TERMS];
    uint64_t cost = 0;  // BufferedUnionScorer::cost = sum of the lists' costs (doc freqs)
  };
  Clause cl[TQ_MAX_TERMS];
  uint32_t n_cl = 0;
  bool empty = false;
  for (uint32_t i = 0; i < q.n_terms; ++i) {
    if (q.occurs[i] > TQ_MUST_NOT) return fail(TQ_ERR_INVALID, "query %u: bad occur", qi);
    const uint32_t id = q.clause_of ? q.clause_of[i] : i;
    uint32_t c = 0;
    while (c < n_cl && cl[c].id != id) ++c;
    if (c == n_cl) {
      cl[n_cl].id = id;
      cl[n_cl].occur = q.occurs[i];
      ++n_cl;
    } else if (cl[c].occur != q.occurs[i]) {
      return fail(TQ_ERR_INVALID, "query %u: clause %u mixes occurs", qi, id);
    }
    if (q.terms[i] == TQ_TERM_ABSENT) continue;
    cl[c].terms[cl[c].n++] = i;
    cl[c].cost += s->terms[q.terms[i]].doc_freq;
  }
  uint32_t must[TQ_MAX_TERMS], should[TQ_MAX_TERMS], mustnot[TQ_MAX_TERMS];
  uint32_t n_must = 0, n_should = 0, n_not = 0;
  for (uint32_t c = 0; c < n_cl; ++c) {
    if (cl[c].occur == TQ_MUST) {
      if (cl[c].n == 0) empty = true;
      must[n_must++] = c;
    } else if (cl[c].n) {
      if (cl[c].occur == TQ_SHOULD)
        should[n_should++] = c;
      else
        mustnot[n_not++] = c;
    }
  }
  // minimum_number_should_match (:272-305): more than there are Should clauses matches
  // nothing; all of them turns them into Must clauses; 1 makes the union required
  uint32_t msm = q.min_should_match;
  if (msm > n_should) empty = true;
  if (!empty && msm >= 2 && msm == n_should) {
    for (uint32_t i = 0; i < n_should; ++i) must[n_must++] = should[i];
    n_should = 0;
    msm = 0;
  }
  if (msm >= 2)
    for (uint32_t i = 0; i < n_should; ++i)
      if (cl[should[i]].n > 1)
        return fail(TQ_ERR_UNSUPPORTED,
                    "query %u: min_should_match > 1 over nested unions stays on the CPU", qi);
  // MustNot clauses only: no include scorer, EmptyScorer (boolean_weight.rs:340-349)
  if (n_must == 0 && n_should == 0) empty = true;
  if (!empty) {
    uint32_t n = 0;
    auto put = [&](uint32_t i, uint32_t role) {
      dq.term[n] = q.terms[i];
      dq.weight[n] = role == TQD_ROLE_MUST_NOT ? 0.0f : q.weights[i];
      dq.roles |= role << (2u * n);
      qbytes += s->terms[q.terms[i]].postings_len;
      ++n;
    };
    auto put_by_weight = [&](uint32_t *idx, uint32_t cnt, uint32_t role) {
      small_stable_sort(idx, idx + cnt,
                       [&](uint32_t a, uint32_t b) { return q.weights[a] > q.weights[b]; });
      for (uint32_t i = 0; i < cnt; ++i) put(idx[i], role);
    };
    uint32_t flat[TQ_MAX_TERMS], n_flat = 0;
    if (n_must) {
      // Must clauses by cost ascending (intersect_scorers, intersection.rs:31): the cheapest
      // leads; then the MustNot terms (they only exclude: densest first), then the Should
      // terms in clause order
      small_stable_sort(must, must + n_must,
                       [&](uint32_t a, uint32_t b) { return cl[a].cost < cl[b].cost; });
      // optional Should lists lead too (MaxScore for RequiredOptionalScorer, see union_body)
      // (only when pruning: with every match scored the extra ownership probes cost 60 %)
      const bool opt_lead = n_should > 0 && msm == 0 && !exhaustive;
      if (opt_lead) {
        for (uint32_t c = 0; c < n_should; ++c)
          for (uint32_t i = 0; i < cl[should[c]].n; ++i) flat[n_flat++] = cl[should[c]].terms[i];
        put_by_weight(flat, n_flat, TQD_ROLE_SHOULD);
        dq.n_opt_lead = n;
        n_flat = 0;
        n_should = 0;
      }
      Clause &lead = cl[must[0]];
      put_by_weight(lead.terms, lead.n, TQD_ROLE_MUST);
      dq.n_lead = n;
      for (uint32_t c = 1; c < n_must; ++c) {
        for (uint32_t i = 0; i < cl[must[c]].n; ++i) put(cl[must[c]].terms[i], TQD_ROLE_MUST);
        dq.clause_end |= 1u << (n - 1u);
      }
      for (uint32_t c = 0; c < n_not; ++c)
        for (uint32_t i = 0; i < cl[mustnot[c]].n; ++i) flat[n_flat++] = cl[mustnot[c]].terms[i];
      small_stable_sort(flat, flat + n_flat, [&](uint32_t a, uint32_t b) {
        return s->terms[q.terms[a]].doc_freq > s->terms[q.terms[b]].doc_freq;
      });
      for (uint32_t i = 0; i < n_flat; ++i) put(flat[i], TQD_ROLE_MUST_NOT);
      for (uint32_t c = 0; c < n_should; ++c)
        for (uint32_t i = 0; i < cl[should[c]].n; ++i) put(cl[should[c]].terms[i], TQD_ROLE_SHOULD);
    } else {
      // no Must: the Should terms form the leading union (by weight descending, as the pure
      // union does), MustNot terms exclude
      for (uint32_t c = 0; c < n_should; ++c)
        for (uint32_t i = 0; i < cl[should[c]].n; ++i) flat[n_flat++] = cl[should[c]].terms[i];
      put_by_weight(flat, n_flat, TQD_ROLE_SHOULD);
      dq.n_lead = n;
      for (uint32_t c = 0; c < n_not; ++c)
        for (uint32_t i = 0; i < cl[mustnot[c]].n; ++i) put(cl[mustnot[c]].terms[i], TQD_ROLE_MUST_NOT);
    }
    dq.min_should = msm;
    dq.n_terms = n;
    bool nonneg = true;
    uint32_t sparse = 0;
    for (uint32_t i = 0; i < n; ++i) {
      nonneg = nonneg && dq.weight[i] >= 0.0f;
      if (!(s->terms[dq.term[i]].dense_blob && s->opt.use_dense)) ++sparse;
    }
    if (!exhaustive && nonneg) {
      dq.flags |= TQD_QF_PRUNE;
      if (q.k <= 2 * TQD_THR_SLOTS) {
        dq.thr_index = n_thr_rows;
        n_thr_rows += 4u;  // union kernel: 64 slots for k <= 16, 256 above; the window kernel 64 / 128
      }
    }
    const uint32_t c_lb = 1u + n + 8u * sparse;
    static const uint32_t kBoolTileNum = std::max<uint32_t>(1u, tune_u32("TQ_BOOL_TILE_NUM", TQD_AND_TILE * 2u));
    dq.tile_blocks = std::min<uint32_t>(TQD_AND_TILE, std::max<uint32_t>(1u, kBoolTileNum / c_lb));
    tile_cost = dq.tile_blocks * c_lb;
    uint32_t acc_tiles = 0;
    for (uint32_t i = 0; i <= TQ_MAX_TERMS; ++i) {
      dq.lead_tile_start[i] = acc_tiles;
      if (i < dq.n_lead)
        acc_tiles += (s->terms[dq.term[i]].n_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
    }
    n_tiles = acc_tiles;
  }
  return TQ_OK;
}


}  // namespace

namespace {

// Per-call execution options (tq_search_opts resolved against the segment's defaults): nothing
// a call needs is read from mutable segment state after this point.
struct CallOpts {
  bool exhaustive;
  float bound_slack;
};

int search_batch_impl(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                      uint32_t out_stride, float *d_out_scores, uint32_t *d_out_docs,
                      uint32_t *d_out_counts, void *hip_stream, const CallOpts &co) {
  if (!s || (!queries && n_queries) || !d_out_scores || !d_out_docs || !d_out_counts)
    return fail(TQ_ERR_INVALID, "tq_search_batch: null argument");
  if (n_queries == 0) return TQ_OK;
  static const bool trace = getenv("TQ_TRACE") != nullptr;
  const auto tr0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->stream;
  int rc = sync_terms(s, st);
  if (rc != TQ_OK) return rc;
  const int opt_exhaustive = co.exhaustive ? 1 : 0;

  // ---- plan
  const bool or_windows_opt = s->opt.or_windows < 0 ? opt_exhaustive != 0 : s->opt.or_windows != 0;
  // launch groups: AND queries whose non-leader lists all have a bitmap run a leaner kernel
  // instantiation (no seek / block-search code: fewer registers, less LDS, more waves per CU)
  // boolean queries (clauses with roles) run the candidate-driven union kernel's BOOL instantiation
  constexpr int kGroups = kNGroups, kAndGeneral = 3, kBool = 4, kShare = 5, kPhSweep = 6, kDense = 7, kAShare = 8;
  // AND queries whose other lists all have a bitmap + byte-wide tfs, pruned, k <= 128, <= 8 lists, on a
  // segment with a doc matrix, whose leader (rarest list) leads at least kAShareMin such queries of the
  // batch: the shared-intersection launch (leader-major, tq_ashare.hip).  TQ_ASHARE=0: the per-query kernel
  static const bool kUseAShare = tune_u32("TQ_ASHARE", 1) != 0;
  static const uint32_t kAShareMin = std::max<uint32_t>(1u, tune_u32("TQ_AS_MIN_LEADS", 4));
  // phrases whose lists ALL have a bitmap, byte-wide tfs and a position directory, the rarest one
  // still about a posting per bitmap word: the bitmap-AND sweep (phrase_sweep_kernel)
  static const uint32_t kPhSweepRatio = tune_u32("TQ_PH_SWEEP_RATIO", 64);  // 0 = never
  // pure unions, pruned, k <= 128, <= 8 terms, on a segment with a doc matrix: the shared-union
  // launch (term-major, tq_ushare.hip); everything else keeps the per-query union kernels
  static const bool kUseShare = tune_u32("TQ_USHARE", 1) != 0;
  // pure unions, NOT pruned, k <= 128, <= 8 terms, positive weights, whose lists together hold at
  // least 1/kDenseRatio of the segment: the doc-major launch (tq_xunion.hip) — up to 256 distinct
  // lists and 8192 queries per batch, one Bm25Weight cache; the rest keeps the window kernel
  static const uint32_t kDenseRatioEnv = tune_u32("TQ_XU_RATIO", 0xFFFFFFFFu);  // (experiments: overrides the option)
  static const uint32_t kDenseMinEnv = tune_u32("TQ_XU_MIN_QUERIES", 0xFFFFFFFFu);
  const uint64_t kDenseRatio = kDenseRatioEnv != 0xFFFFFFFFu ? kDenseRatioEnv : (uint32_t)s->opt.xunion_ratio;
  const uint32_t kDenseMinQueries = std::max<uint32_t>(1u, kDenseMinEnv != 0xFFFFFFFFu ? kDenseMinEnv : (uint32_t)s->opt.xunion_min_queries);
  uint32_t dense_cache = 0xFFFFFFFFu;
  if (s->share_span_terms != s->terms.size()) {  // (terms are prepared rarely)
    uint64_t lo = ~0ull, hi = 0;
    for (const TermHost &th : s->terms)
      for (const void *ptr : {th.dense_blob, th.tf8_blob})
        if (ptr) {
          lo = std::min<uint64_t>(lo, (uint64_t)ptr);
          hi = std::max<uint64_t>(hi, (uint64_t)ptr);
        }
    if (lo == ~0ull) lo = hi = 8;
    s->share_table_lo = lo - 8;
    s->share_span_ok = hi - s->share_table_lo < (8ull << 32);
    s->share_span_terms = s->terms.size();
  }
  if (!s->plan) s->plan = new PlanScratch();
  Group(&groups)[kGroups] = s->plan->groups;
  for (Group &g : groups) g.reset();
  groups[kBool].mode = TQ_MODE_OR;
  groups[kShare].mode = TQ_MODE_OR;
  groups[kPhSweep].mode = TQ_MODE_PHRASE;
  groups[kDense].mode = TQ_MODE_OR;
  groups[kAShare].mode = TQ_MODE_AND;
  s->plan->xrow_term.clear();
  s->plan->xrow_of.clear();
  groups[0].mode = TQ_MODE_AND;
  groups[1].mode = TQ_MODE_OR;
  groups[2].mode = TQ_MODE_PHRASE;
  groups[kAndGeneral].mode = TQ_MODE_AND;
  std::vector<const float *> caches;
  uint64_t algo_bytes = 0;
  uint32_t n_thr_rows = 0;
  bool phrase_all_dense = true;
  // which Bm25Weight cache every query uses (pointer identity; a handful per batch)
  PlanScratch &ps_plan = *s->plan;
  ps_plan.q_cache.resize(n_queries);
  for (uint32_t qi = 0; qi < n_queries; ++qi) {
    const float *tc = queries[qi].tf_cache;
    uint32_t cache_idx = 0;
    if (tc) {
      for (; cache_idx < caches.size(); ++cache_idx)
        if (caches[cache_idx] == tc) break;
      if (cache_idx == caches.size()) caches.push_back(tc);
    }
    ps_plan.q_cache[qi] = cache_idx;
  }
  // Which list would lead an AND query in the shared-intersection launch (0xFFFFFFFF: the query does not
  // qualify), and how many queries of the batch every list would lead; each distinct list's bytes once
  // (tq_batch_stats.unique_bytes: what the batch needs from the index when nothing is read twice).
  const bool ashare_on = kUseAShare && !opt_exhaustive && s->d_docmat && s->opt.use_dense && s->share_span_ok;
  auto ashare_leader = [&](const tq_query &q, uint32_t cache_idx) -> uint32_t {
    if (q.mode != TQ_MODE_AND || q.n_terms < 2 || q.n_terms > TQD_AS_MAX_TERMS || q.k == 0 || q.k > 128u ||
        !q.terms || !q.weights || cache_idx >= 256u)
      return 0xFFFFFFFFu;
    uint32_t best = 0xFFFFFFFFu, best_i = 0;
    for (uint32_t i = 0; i < q.n_terms; ++i) {
      if (q.terms[i] == TQ_TERM_ABSENT || q.terms[i] >= s->terms.size() || !(q.weights[i] >= 0.0f)) return 0xFFFFFFFFu;
      const uint32_t df = s->terms[q.terms[i]].doc_freq;
      if (df < best) {  // (first of the rarest: what the stable sort by doc freq puts in front)
        best = df;
        best_i = i;
      }
    }
    for (uint32_t i = 0; i < q.n_terms; ++i) {
      if (i == best_i) continue;
      const TermHost &th = s->terms[q.terms[i]];
      if (!(th.dense_blob && th.tf8_blob)) return 0xFFFFFFFFu;
    }
    return q.terms[best_i];
  };
  uint64_t unique_bytes = 0;
  {
    PlanScratch &ps = ps_plan;
    if (ps.term_stamp.size() < 2 * s->terms.size()) ps.term_stamp.resize(2 * s->terms.size(), 0u);
    if (++ps.batch_stamp == 0u) {
      std::fill(ps.term_stamp.begin(), ps.term_stamp.end(), 0u);
      ps.batch_stamp = 1u;
    }
    ps.and_lead_count.assign(ashare_on ? s->terms.size() : 0, 0u);
    ps.q_leader.resize(ashare_on ? n_queries : 0);
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const tq_query &q = queries[qi];
      if (!q.terms || q.n_terms > TQ_MAX_TERMS) continue;  // (reported by plan_query)
      for (uint32_t i = 0; i < q.n_terms; ++i) {
        const uint32_t h = q.terms[i];
        if (h >= s->terms.size()) continue;
        if (ps.term_stamp[2 * h] != ps.batch_stamp) {
          ps.term_stamp[2 * h] = ps.batch_stamp;
          unique_bytes += s->terms[h].postings_len;
        }
        if (q.mode == TQ_MODE_PHRASE && ps.term_stamp[2 * h + 1] != ps.batch_stamp) {
          ps.term_stamp[2 * h + 1] = ps.batch_stamp;
          unique_bytes += s->terms[h].positions_len;
        }
      }
      if (ashare_on) {
        const uint32_t lh = ashare_leader(q, ps.q_cache[qi]);
        ps.q_leader[qi] = lh;
        if (lh != 0xFFFFFFFFu) ++ps.and_lead_count[lh];
      }
    }
  }
  // One query -> its descriptor in its launch group.  Reads the segment and the caller's query only,
  // writes to the groups / counters it is handed: large pruned batches are planned in slabs of
  // queries by the planner's threads, each into its own groups, which are then laid end to end.
  auto plan_query = [&](uint32_t qi, Group *groups, uint32_t &n_thr_rows, uint64_t &algo_bytes,
                        bool &phrase_all_dense) -> int {
    const tq_query &q = queries[qi];
    if (q.n_terms == 0 || q.n_terms > TQ_MAX_TERMS)
      return fail(TQ_ERR_INVALID, "query %u: n_terms %u not in 1..%u", qi, q.n_terms, TQ_MAX_TERMS);
    if (q.k == 0 || q.k > TQ_MAX_K || q.k > out_stride)
      return fail(TQ_ERR_INVALID, "query %u: k %u not in 1..min(%u, out_stride %u)", qi, q.k,
                  TQ_MAX_K, out_stride);
    if (!q.terms || !q.weights || !q.tf_cache)
      return fail(TQ_ERR_INVALID, "query %u: null terms/weights/tf_cache", qi);
    if (q.mode > TQ_MODE_BOOL) return fail(TQ_ERR_INVALID, "query %u: bad mode", qi);
    if (q.mode == TQ_MODE_BOOL && !q.occurs)
      return fail(TQ_ERR_INVALID, "query %u: TQ_MODE_BOOL needs occurs", qi);
    if (q.mode == TQ_MODE_PHRASE && (q.n_terms < 2 || !q.phrase_offsets))
      return fail(TQ_ERR_INVALID, "query %u: a phrase needs >= 2 terms and offsets", qi);
    if (q.mode == TQ_MODE_PHRASE && q.n_terms > 8)
      return fail(TQ_ERR_UNSUPPORTED, "query %u: device phrases take at most 8 terms", qi);
    // NaN / inf weights (boosts) would break the total order of the top-k keys and of merge_top_k
    for (uint32_t i = 0; i < (q.mode == TQ_MODE_PHRASE ? 1u : q.n_terms); ++i)
      if (!std::isfinite(q.weights[i]))
        return fail(TQ_ERR_INVALID, "query %u: weight %u is not finite", qi, i);
    const uint32_t cache_idx = ps_plan.q_cache[qi];

    TqdQuery dq{};
    dq.thr_index = 0xFFFFFFFFu;
    dq.k = q.k;
    dq.cache_idx = cache_idx;
    dq.mode = q.mode;
    bool any_absent = false;
    for (uint32_t i = 0; i < q.n_terms; ++i) {
      if (q.terms[i] == TQ_TERM_ABSENT) {
        any_absent = true;
        continue;
      }
      if (q.terms[i] >= s->terms.size())
        return fail(TQ_ERR_INVALID, "query %u: unknown term handle %u", qi, q.terms[i]);
    }
    int mode = q.mode;
    bool ph_sweep = false, ashare = false;
    uint32_t n_tiles = 0, tile_cost = 1;
    bool all_dense = true;
    uint64_t qbytes = 8ull * q.k;
    if (mode == TQ_MODE_AND || mode == TQ_MODE_PHRASE) {
      if (!any_absent) {
        // stable sort by doc_freq asc (block_wand_intersection.rs:26-29 / intersection.rs:93)
        uint32_t order[TQ_MAX_TERMS];
        for (uint32_t i = 0; i < q.n_terms; ++i) order[i] = i;
        small_stable_sort(order, order + q.n_terms, [&](uint32_t a, uint32_t b) {
          return s->terms[q.terms[a]].doc_freq < s->terms[q.terms[b]].doc_freq;
        });
        uint32_t max_off = 0;
        if (mode == TQ_MODE_PHRASE)
          for (uint32_t i = 0; i < q.n_terms; ++i) max_off = std::max(max_off, q.phrase_offsets[i]);
        for (uint32_t i = 0; i < q.n_terms; ++i) {
          const uint32_t src = order[i];
          dq.term[i] = q.terms[src];
          dq.weight[i] = mode == TQ_MODE_PHRASE ? q.weights[0] : q.weights[src];
          if (mode == TQ_MODE_PHRASE) dq.phrase_off[i] = max_off - q.phrase_offsets[src];
          qbytes += s->terms[q.terms[src]].postings_len;
          if (mode == TQ_MODE_PHRASE) {
            if (s->terms[q.terms[src]].positions_len == 0)
              return fail(TQ_ERR_UNSUPPORTED, "query %u: phrase on a field without positions", qi);
            qbytes += s->terms[q.terms[src]].positions_len;
          }
        }
        dq.n_terms = q.n_terms;
        if (mode == TQ_MODE_AND && q.n_terms == 1) {
          mode = TQ_MODE_OR;  // TermWeight::for_each_pruning: every doc of the list
        } else if (mode == TQ_MODE_AND) {
          // cost of one leader block: its own decode + the distinct blocks of the non-dense
          // lists its 128 candidates can fall into (each decoded by the whole wave, serially)
          const uint32_t lead_blocks = s->terms[dq.term[0]].n_blocks;
          uint32_t c_lb = 1;
          for (uint32_t i = 1; i < q.n_terms; ++i) {
            const TermHost &th = s->terms[dq.term[i]];
            if (th.dense_blob && s->opt.use_dense) continue;
            all_dense = false;
            c_lb += 2u * std::min<uint32_t>(128u, (th.n_blocks + lead_blocks - 1) / lead_blocks);
          }
          static const uint32_t kAndTileNum = std::max<uint32_t>(1u, tune_u32("TQ_AND_TILE_NUM", TQD_AND_TILE));
          dq.tile_blocks = std::min<uint32_t>(TQD_AND_TILE, std::max<uint32_t>(1u, kAndTileNum / c_lb));
          tile_cost = dq.tile_blocks * c_lb;
          n_tiles = (lead_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
          bool nonneg = true;
          for (uint32_t i = 0; i < q.n_terms; ++i) nonneg = nonneg && dq.weight[i] >= 0.0f;
          if (ashare_on && nonneg && all_dense) {
            const uint32_t lh = ps_plan.q_leader[qi];
            ashare = lh != 0xFFFFFFFFu && lh == dq.term[0] && ps_plan.and_lead_count[lh] >= kAShareMin;
          }
          if (ashare) {  // (planned per leader, not per query: build_ashare_plan)
            dq.flags |= TQD_QF_PRUNE;
            dq.thr_index = n_thr_rows;
            n_thr_rows += q.k <= 16u ? 1u : 4u;  // 64 hashed score slots for k <= 16, 256 above
            n_tiles = 0;
          } else if (!opt_exhaustive && nonneg) {  // block-max bounds need weights >= 0
            dq.flags |= TQD_QF_PRUNE;
            // the shared threshold pays off on long lists only; k-th largest of 64 slots needs k <= 64
            if (q.k <= TQD_THR_SLOTS && n_tiles >= 2) dq.thr_index = n_thr_rows++;
          }
        } else {  // phrase: leader-block tiles like AND; every match also walks its positions
          const uint32_t lead_blocks = s->terms[dq.term[0]].n_blocks;
          ph_sweep = kPhSweepRatio && q.n_terms <= 4u && s->opt.use_dense &&
                     (uint64_t)s->terms[dq.term[0]].doc_freq * kPhSweepRatio >= s->max_doc;
          for (uint32_t i = 0; ph_sweep && i < q.n_terms; ++i) {
            const TermHost &th = s->terms[dq.term[i]];
            if (!(th.dense_blob && th.tf8_blob && th.posdir_blob)) ph_sweep = false;
          }
          // the lean instantiation needs a bitmap, a doc-matrix column and a position directory
          // for every non-leader list
          for (uint32_t i = 1; !ph_sweep && i < q.n_terms; ++i) {
            const TermHost &th = s->terms[dq.term[i]];
            const bool col = ((s->h_dterms[dq.term[i]].has_freq >> 8) & 0xFFu) != 0u;
            if (!(th.dense_blob && th.posdir_blob && th.tf8_blob && col && s->opt.use_dense && s->d_docmat))
              phrase_all_dense = false;
          }
          // (64-block tiles: one leader block per lane of the pre-filter; 32 was 10 % slower)
          static const uint32_t kPhTile = std::min<uint32_t>(TQD_AND_TILE, std::max<uint32_t>(1u, tune_u32("TQ_PH_TILE_BLOCKS", 64)));
          dq.tile_blocks = kPhTile;
          tile_cost = 2u * kPhTile;
          n_tiles = (lead_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
          if (ph_sweep) {  // tiles are runs of 2048 bitmap words (phrase_sweep_kernel's SWEEP_WORDS)
            const uint32_t n_words = (s->max_doc + 31u) / 32u;
            n_tiles = (n_words + 2047u) / 2048u;
            tile_cost = 64u;
          }
        }
      }
    }
    bool bool_done = false, share = false, dense_u = false;
    if (q.mode == TQ_MODE_BOOL) {
      const int rc = plan_bool_query(s, q, qi, dq, qbytes, n_tiles, tile_cost, n_thr_rows, opt_exhaustive != 0);
      if (rc != TQ_OK) return rc;
      mode = TQ_MODE_OR;  // runs in the union launch group
      bool_done = true;
    }
    if (mode == TQ_MODE_OR && !bool_done) {
      if (q.mode == TQ_MODE_OR) {
        uint32_t n = 0;
        for (uint32_t i = 0; i < q.n_terms; ++i) {
          if (q.terms[i] == TQ_TERM_ABSENT) continue;
          dq.term[n] = q.terms[i];
          dq.weight[n] = q.weights[i];
          qbytes += s->terms[q.terms[i]].postings_len;
          ++n;
        }
        dq.n_terms = n;
      }
      // terms by weight descending (stable): the score sum order of the union kernel, and what
      // makes the low-weight (dense) lists the non-essential suffix of MaxScore pruning
      {
        uint32_t order[TQ_MAX_TERMS];
        for (uint32_t i = 0; i < dq.n_terms; ++i) order[i] = i;
        small_stable_sort(order, order + dq.n_terms,
                         [&](uint32_t a, uint32_t b) { return dq.weight[a] > dq.weight[b]; });
        uint32_t t2[TQ_MAX_TERMS];
        float w2[TQ_MAX_TERMS];
        for (uint32_t i = 0; i < dq.n_terms; ++i) {
          t2[i] = dq.term[order[i]];
          w2[i] = dq.weight[order[i]];
        }
        bool nonneg = true;
        for (uint32_t i = 0; i < dq.n_terms; ++i) {
          dq.term[i] = t2[i];
          dq.weight[i] = w2[i];
          nonneg = nonneg && w2[i] >= 0.0f;
        }
        if (!opt_exhaustive && nonneg && dq.n_terms) {
          dq.flags |= TQD_QF_PRUNE;
          if (q.k <= 2 * TQD_THR_SLOTS) {  // k-th largest of 64 (128) slots needs k <= 64 (128)
            dq.thr_index = n_thr_rows;
            n_thr_rows += 4u;  // union kernel: 64 slots for k <= 16, 256 above; the window kernel 64 / 128
          }
        }
      }
      share = kUseShare && s->share_span_ok && !or_windows_opt && (dq.flags & TQD_QF_PRUNE) && dq.thr_index != 0xFFFFFFFFu &&
              dq.n_terms >= 1 && dq.n_terms <= TQD_US_MAX_TERMS && s->d_docmat && s->opt.use_dense &&
              cache_idx < 256u;
      for (uint32_t i = 0; share && i < dq.n_terms; ++i)  // (lists with a bitmap carry byte-wide tfs)
        if (s->terms[dq.term[i]].dense_blob && !s->terms[dq.term[i]].tf8_blob) share = false;
      if (share) {
        // (planned per term, not per query: build_share_plan)
      } else if (or_windows_opt) {
        uint32_t max_last = 0;
        for (uint32_t i = 0; i < dq.n_terms; ++i)
          max_last = std::max(max_last, s->terms[dq.term[i]].last_doc);
        if (dq.n_terms) n_tiles = max_last / TQD_OR_WINDOW + 1;
        // the doc-major launch?
        PlanScratch &ps = *s->plan;
        dense_u = kDenseRatio && opt_exhaustive && s->opt.use_dense && dq.n_terms >= 1 && dq.n_terms <= 8 &&
                  q.k <= 128 && (dense_cache == 0xFFFFFFFFu || dense_cache == cache_idx) &&
                  groups[kDense].queries.size() < TQK_XU_MAX_QUERIES;
        uint64_t sum_df = 0;
        uint32_t new_rows = 0;
        for (uint32_t i = 0; dense_u && i < dq.n_terms; ++i) {
          if (!(dq.weight[i] > 0.0f)) dense_u = false;
          sum_df += s->terms[dq.term[i]].doc_freq;
          bool seen = ps.xrow_of.count(xrow_key(dq.term[i], dq.weight[i])) != 0;
          for (uint32_t j = 0; j < i; ++j) seen = seen || (dq.term[j] == dq.term[i] && dq.weight[j] == dq.weight[i]);
          if (!seen) ++new_rows;
        }
        if (dense_u && (sum_df * kDenseRatio < s->max_doc || ps.xrow_term.size() + new_rows > TQK_XU_MAX_ROWS - 1u))
          dense_u = false;
        for (uint32_t i = 0; dense_u && i < dq.n_terms; ++i) {
          const TermHost &th = s->terms[dq.term[i]];
          if (th.dense_blob && th.tf8_blob) continue;
          bool ok = false;
          const int frc = build_flat(s, dq.term[i], st, &ok);
          if (frc != TQ_OK) return frc;
          if (!ok) dense_u = false;
        }
        if (dense_u) {
          dense_cache = cache_idx;
          for (uint32_t i = 0; i < dq.n_terms; ++i) {
            const uint64_t key = xrow_key(dq.term[i], dq.weight[i]);
            if (ps.xrow_of.emplace(key, (uint32_t)ps.xrow_term.size()).second) ps.xrow_term.push_back(key);
          }
          dq.thr_index = n_thr_rows;
          n_thr_rows += 4u;
        }
      } else if (dq.n_terms) {
        // candidate-driven: every list leads its own run of tiles; a candidate probes all the
        // other lists (non-dense ones cost a seek + a block search)
        uint32_t sparse = 0;
        for (uint32_t i = 0; i < dq.n_terms; ++i)
          if (!(s->terms[dq.term[i]].dense_blob && s->opt.use_dense)) ++sparse;
        const uint32_t c_lb = 1u + dq.n_terms + 8u * sparse;
        static const uint32_t kOrTileBlocks = tune_u32("TQ_OR_TILE_BLOCKS", 0);
        static const uint32_t kOrTileNum = std::max<uint32_t>(1u, tune_u32("TQ_OR_TILE_NUM", TQD_AND_TILE * 2u));
        dq.tile_blocks = kOrTileBlocks ? std::min<uint32_t>(kOrTileBlocks, TQD_AND_TILE)
                                       : std::min<uint32_t>(TQD_AND_TILE, std::max<uint32_t>(1u, kOrTileNum / c_lb));
        tile_cost = dq.tile_blocks * c_lb;
        uint32_t acc_tiles = 0;
        for (uint32_t i = 0; i < dq.n_terms; ++i) {
          dq.lead_tile_start[i] = acc_tiles;
          acc_tiles += (s->terms[dq.term[i]].n_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
        }
        for (uint32_t i = dq.n_terms; i <= TQ_MAX_TERMS; ++i) dq.lead_tile_start[i] = acc_tiles;
        dq.n_lead = dq.n_terms;
        n_tiles = acc_tiles;
      }
    }
    algo_bytes += qbytes;
    dq.n_tiles = n_tiles;
    Group &g = groups[bool_done ? kBool : (share ? kShare : (dense_u ? kDense : (ph_sweep ? kPhSweep : (ashare ? kAShare : ((mode == TQ_MODE_AND && !all_dense) ? kAndGeneral : mode)))))];
    dq.mode = (uint32_t)mode;
    g.queries.push_back(dq);
    g.tile_cost.push_back(tile_cost);
    g.out_index.push_back(qi);
    g.max_k = std::max(g.max_k, q.k);
    return TQ_OK;
  };
  const auto tr0a = std::chrono::steady_clock::now();  // (after validation of the context, the cache table and the pre-pass)
  static const uint32_t kQuerySlabMin = tune_u32("TQ_PLAN_QUERY_PAR_MIN", 4096);
  const uint32_t q_slabs = (!opt_exhaustive && n_queries >= kQuerySlabMin) ? std::min<uint32_t>(plan_threads(), 8u) : 1u;
  if (q_slabs <= 1) {
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const int qrc = plan_query(qi, groups, n_thr_rows, algo_bytes, phrase_all_dense);
      if (qrc != TQ_OK) return qrc;
    }
  } else {
    std::vector<QuerySlab> &qs = ps_plan.q_slabs;
    if (qs.size() < q_slabs) qs.resize(q_slabs);
    parallel_slabs(q_slabs, [&](uint32_t sb) {
      QuerySlab &Q = qs[sb];
      for (int gi = 0; gi < kGroups; ++gi) {
        Q.groups[gi].reset();
        Q.groups[gi].mode = groups[gi].mode;
      }
      Q.n_thr_rows = 0;
      Q.algo_bytes = 0;
      Q.phrase_all_dense = true;
      Q.rc = TQ_OK;
      const uint32_t q0 = (uint32_t)((uint64_t)n_queries * sb / q_slabs), q1 = (uint32_t)((uint64_t)n_queries * (sb + 1) / q_slabs);
      for (uint32_t qi = q0; qi < q1; ++qi) {
        Q.rc = plan_query(qi, Q.groups, Q.n_thr_rows, Q.algo_bytes, Q.phrase_all_dense);
        if (Q.rc != TQ_OK) {
          Q.err = g_last_error;  // (this thread's slot: handed to the caller's below)
          break;
        }
      }
    });
    uint32_t thr_base[9] = {0};
    size_t g_base[kNGroups][9] = {};
    for (uint32_t sb = 0; sb < q_slabs; ++sb) {
      if (qs[sb].rc != TQ_OK) {
        g_last_error = qs[sb].err;
        return qs[sb].rc;
      }
      thr_base[sb + 1] = thr_base[sb] + qs[sb].n_thr_rows;
      algo_bytes += qs[sb].algo_bytes;
      phrase_all_dense = phrase_all_dense && qs[sb].phrase_all_dense;
      for (int gi = 0; gi < kGroups; ++gi) g_base[gi][sb + 1] = g_base[gi][sb] + qs[sb].groups[gi].queries.size();
    }
    n_thr_rows = thr_base[q_slabs];
    for (int gi = 0; gi < kGroups; ++gi) {
      Group &g = groups[gi];
      const size_t total = g_base[gi][q_slabs];
      g.queries.resize(total);
      g.tile_cost.resize(total);
      g.out_index.resize(total);
      for (uint32_t sb = 0; sb < q_slabs; ++sb) g.max_k = std::max(g.max_k, qs[sb].groups[gi].max_k);
    }
    parallel_slabs(q_slabs, [&](uint32_t sb) {  // slab order = query order inside every group
      for (int gi = 0; gi < kGroups; ++gi) {
        const Group &src = qs[sb].groups[gi];
        Group &g = groups[gi];
        const size_t at = g_base[gi][sb], n = src.queries.size();
        for (size_t i = 0; i < n; ++i) {
          g.queries[at + i] = src.queries[i];
          if (g.queries[at + i].thr_index != 0xFFFFFFFFu) g.queries[at + i].thr_index += thr_base[sb];
        }
        if (n) {
          memcpy(g.tile_cost.data() + at, src.tile_cost.data(), n * sizeof(uint32_t));
          memcpy(g.out_index.data() + at, src.out_index.data(), n * sizeof(uint32_t));
        }
      }
    });
  }
  const auto tr0b = std::chrono::steady_clock::now();
  // too few queries to pay for the tile rows: they keep the window kernel
  if (!groups[kDense].queries.empty() && groups[kDense].queries.size() < kDenseMinQueries) {
    Group &d = groups[kDense], &o = groups[1];
    o.queries.append(d.queries.begin(), d.queries.end());
    o.tile_cost.insert(o.tile_cost.end(), d.tile_cost.begin(), d.tile_cost.end());
    o.out_index.insert(o.out_index.end(), d.out_index.begin(), d.out_index.end());
    o.max_k = std::max(o.max_k, d.max_k);
    d.reset();
    d.mode = TQ_MODE_OR;
  }
  // tiles -> chunks -> partial lists
  uint32_t total_parts = 0;
  size_t partial_bytes = 0;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device);
  for (Group &g : groups) {
    if (g.queries.empty()) continue;
    const int crc = &g == &groups[kShare]    ? build_share_plan(s, g, *s->plan)
                    : &g == &groups[kAShare] ? build_ashare_plan(s, g, *s->plan)
                    : &g == &groups[kDense]  ? build_dense_plan(s, g, *s->plan, (uint32_t)std::max(1, cus))
                                             : build_group_chunks(g, or_windows_opt && &g != &groups[kBool], *s->plan,
                                                                  &g == &groups[kBool]);
    if (crc != TQ_OK) return crc;
  }
  // partial lists of all groups share one buffer; its stride is per group (kpl*64 keys)
  size_t part_off_bytes[kGroups] = {};
  for (int gi = 0; gi < kGroups; ++gi) {
    Group &g = groups[gi];
    part_off_bytes[gi] = partial_bytes;
    if (gi == kShare || gi == kDense || gi == kAShare) {  // result lists: part_start / n_parts count 8-byte entries (build_share_plan)
      if (!g.queries.empty())
        partial_bytes += ((size_t)g.queries.back().part_start + g.queries.back().n_parts) * sizeof(uint64_t);
      continue;
    }
    uint32_t parts = 0;
    for (TqdQuery &dq : g.queries) {
      dq.part_start = parts;
      parts += dq.n_parts;
    }
    total_parts += parts;
    partial_bytes += (size_t)parts * (size_t)g.kpl * 64u * sizeof(uint64_t);
  }
  // From here to the event behind the batch's last kernel the device's shared scratch is this batch's.
  DeviceScratch &sc = *s->dscratch;
  std::unique_lock<std::mutex> scratch_lock(sc.m);
  if (!sc.ev_last) HIP_TRY(hipEventCreateWithFlags(&sc.ev_last, hipEventDisableTiming));
  rc = sc.partials.ensure(partial_bytes + 256);
  if (rc == TQ_OK) rc = s->d_qmatches.ensure((size_t)n_queries * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;

  // ---- stage: [caches][per group: queries | tile_starts | out_index]
  size_t stage = 0;
  const size_t o_caches = 0;
  stage += caches.size() * 256 * sizeof(float);
  for (Group &g : groups) {
    if (g.queries.empty()) continue;
    stage = (stage + 15) & ~(size_t)15;
    g.o_queries = stage;
    stage += g.queries.size() * sizeof(TqdQuery);
    stage = (stage + 15) & ~(size_t)15;
    g.o_tiles = stage;
    stage += g.tile_starts.size() * sizeof(uint32_t);
    stage = (stage + 15) & ~(size_t)15;
    g.o_outidx = stage;
    stage += g.out_index.size() * sizeof(uint32_t);
    stage = (stage + 15) & ~(size_t)15;
    g.o_chunks = stage;
    stage += g.chunk_recs.size() * sizeof(uint4);
    stage = (stage + 15) & ~(size_t)15;
    g.o_sinks = stage;
    stage += sizeof(TqkSinks);
    if (&g == &groups[kShare]) {
      stage = (stage + 63) & ~(size_t)63;
      g.o_leads = stage;
      stage += s->plan->leads.size() * sizeof(TqdLead);
      stage = (stage + 15) & ~(size_t)15;
      g.o_tasks = stage;
      stage += s->plan->tasks.size() * sizeof(uint4);
    }
    if (&g == &groups[kAShare]) {
      stage = (stage + 63) & ~(size_t)63;
      g.o_leads = stage;
      stage += s->plan->aleads.size() * sizeof(TqdALead);
      stage = (stage + 15) & ~(size_t)15;
      g.o_tasks = stage;
      stage += s->plan->atasks.size() * sizeof(uint4);
    }
    if (&g == &groups[kDense]) {  // (o_leads: the rows, o_tasks: the queries)
      stage = (stage + 63) & ~(size_t)63;
      g.o_leads = stage;
      stage += s->plan->xrows.size() * sizeof(TqkDenseRow);
      stage = (stage + 15) & ~(size_t)15;
      g.o_tasks = stage;
      stage += s->plan->xqueries.size() * sizeof(TqkDenseQuery);
    }
  }
  const auto tr1 = std::chrono::steady_clock::now();
  if (s->stage_in_flight) {  // (the pinned staging buffer is reused: the previous batch's copy must have left it)
    HIP_TRY(hipEventSynchronize(s->ev_stage_done));
    s->stage_in_flight = false;
  }
  const auto tr1w = std::chrono::steady_clock::now();  // time spent waiting for the GPU is not planning time
  static const bool kCopyStream = tune_u32("TQ_COPY_STREAM", 1) != 0;
  const int bx = kCopyStream ? (int)(s->batches_enqueued & 1u) : 0;
  DevBuf &dstage = bx ? s->d_stage_alt : s->d_stage;
  rc = s->h_stage.ensure(stage);
  if (rc == TQ_OK) rc = s->d_stage.ensure(stage);
  // (both buffers grow with the first batch that needs it: a growth is a hipFree, i.e. a device-wide
  // synchronisation, and must not wait for the second batch of a new workload)
  if (rc == TQ_OK && kCopyStream) rc = s->d_stage_alt.ensure(stage);
  if (rc != TQ_OK) return rc;
  uint8_t *hs = (uint8_t *)s->h_stage.p;
  for (size_t c = 0; c < caches.size(); ++c)
    memcpy(hs + o_caches + c * 256 * sizeof(float), caches[c], 256 * sizeof(float));
  // the two big tables of a group (descriptors, chunk records: megabytes per 10 000-query batch) are
  // copied by the planner's threads, a quarter each
  auto big_copy = [&](uint8_t *dst, const void *src, size_t bytes) {
    const uint32_t parts = bytes >= (1u << 20) ? std::min<uint32_t>(plan_threads(), 4u) : 1u;
    parallel_slabs(parts, [&](uint32_t pi) {
      const size_t a = (bytes * pi / parts) & ~(size_t)63, b = pi + 1 == parts ? bytes : (bytes * (pi + 1) / parts) & ~(size_t)63;
      memcpy(dst + a, (const uint8_t *)src + a, b - a);
    });
  };
  for (Group &g : groups) {
    if (g.queries.empty()) continue;
    big_copy(hs + g.o_queries, g.queries.data(), g.queries.size() * sizeof(TqdQuery));
    memcpy(hs + g.o_tiles, g.tile_starts.data(), g.tile_starts.size() * sizeof(uint32_t));
    memcpy(hs + g.o_outidx, g.out_index.data(), g.out_index.size() * sizeof(uint32_t));
    big_copy(hs + g.o_chunks, g.chunk_recs.data(), g.chunk_recs.size() * sizeof(uint4));
    if (&g == &groups[kShare]) {
      big_copy(hs + g.o_leads, s->plan->leads.data(), s->plan->leads.size() * sizeof(TqdLead));
      memcpy(hs + g.o_tasks, s->plan->tasks.data(), s->plan->tasks.size() * sizeof(uint4));
    }
    if (&g == &groups[kAShare]) {
      memcpy(hs + g.o_leads, s->plan->aleads.data(), s->plan->aleads.size() * sizeof(TqdALead));
      big_copy(hs + g.o_tasks, s->plan->atasks.data(), s->plan->atasks.size() * sizeof(uint4));
    }
    if (&g == &groups[kDense]) {
      memcpy(hs + g.o_leads, s->plan->xrows.data(), s->plan->xrows.size() * sizeof(TqkDenseRow));
      memcpy(hs + g.o_tasks, s->plan->xqueries.data(), s->plan->xqueries.size() * sizeof(TqkDenseQuery));
    }
  }
  for (int gi = 0; gi < kGroups; ++gi) {
    Group &g = groups[gi];
    if (g.queries.empty()) continue;
    TqkSinks sk{};
    sk.partials = (uint64_t *)((uint8_t *)sc.partials.p + part_off_bytes[gi]);
    sk.match_counter = s->d_match_counter;
    sk.query_matches = (uint32_t *)s->d_qmatches.p;
    sk.out_index = (const uint32_t *)((const uint8_t *)dstage.p + g.o_outidx);
    memcpy(hs + g.o_sinks, &sk, sizeof sk);
  }
  const auto tr2 = std::chrono::steady_clock::now();
  const int slot = (int)(s->batches_timed % tq_segment::kTimingRing);
  // the scratch below is shared with the previous batch: wait for it if it ran on another stream
  rc = order_after_last_batch(s, st);
  if (rc != TQ_OK) return rc;
  // ... and the device's shared scratch with whichever segment's batch used it last
  if (sc.in_flight && sc.last_stream != st) HIP_TRY(hipStreamWaitEvent(st, sc.ev_last, 0));
  if (s->opt.timing) HIP_TRY(hipEventRecord(s->ev_t0[slot], st));
  // From here on work is in flight that reads the staging buffer: a failure below must not let the
  // next call overwrite it under kernels that were already launched (the events that order the
  // buffers are only recorded at the end), so every error return first drains the streams.
  struct DrainOnError {
    tq_segment *s;
    hipStream_t st;
    bool armed = true;
    ~DrainOnError() {
      if (!armed) return;
      if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
      if (s->side_stream) (void)hipStreamSynchronize(s->side_stream);
      (void)hipStreamSynchronize(st);
    }
  } drain_on_error{s, st};
  if (kCopyStream) {
    // buffer bx was last read by the batch before the previous one: the copy waits for that
    // batch's end (recorded on its stream), the kernels below wait for the copy
    if (s->buf_used[bx]) HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_buf_free[bx], 0));
    HIP_TRY(hipMemcpyAsync(dstage.p, hs, stage, hipMemcpyHostToDevice, s->copy_stream));
    HIP_TRY(hipEventRecord(s->ev_stage_done, s->copy_stream));
    HIP_TRY(hipEventRecord(s->ev_copy_done[bx], s->copy_stream));
    HIP_TRY(hipStreamWaitEvent(st, s->ev_copy_done[bx], 0));
  } else {
    HIP_TRY(hipMemcpyAsync(dstage.p, hs, stage, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(s->ev_stage_done, st));
  }
  s->stage_in_flight = true;
  HIP_TRY(hipMemsetAsync(s->d_match_counter, 0, sizeof(unsigned long long), st));
  HIP_TRY(hipMemsetAsync(s->d_qmatches.p, 0, (size_t)n_queries * sizeof(uint32_t), st));
  s->last_batch_queries = n_queries;
  if (n_thr_rows) {
    const size_t thr_bytes = (size_t)n_thr_rows * TQD_THR_SLOTS * sizeof(uint32_t);
    rc = s->d_thr.ensure(thr_bytes);
    if (rc != TQ_OK) return rc;
    // TQ_KEEP_THR=1 (experiments only): the slots keep the previous batch's final values, i.e. the
    // same batch run again starts from its final thresholds (what perfect threshold knowledge buys)
    static const bool kKeepThr = tune_u32("TQ_KEEP_THR", 0) != 0;
    if (!kKeepThr || !s->thr_seeded) HIP_TRY(hipMemsetAsync(s->d_thr.p, 0, thr_bytes, st));
    s->thr_seeded = true;
  }

  // shared-union launch: thr_val | list_count per query, then the task counter (zeroed per batch);
  // staging lists of the persistent grid
  uint32_t share_grid = 0;
  const size_t n_share = groups[kShare].queries.size();
  // (the doc-major launch only exists without pruning, the shared-union launch only with it: the
  // two never meet in one batch and share the per-query words and the staging buffer)
  const size_t n_dense = groups[kDense].queries.size();
  if (n_dense) {
    const size_t words = 2 * n_dense + 16;
    rc = s->d_share_words.ensure(words * sizeof(uint32_t));
    if (rc == TQ_OK)
      rc = sc.share_stage.ensure((size_t)s->plan->xgrid * n_dense * tqk_share_capl(groups[kDense].kpl) * sizeof(uint64_t));
    if (rc != TQ_OK) return rc;
    HIP_TRY(hipMemsetAsync(s->d_share_words.p, 0, words * sizeof(uint32_t), st));
  }
  if (n_share) {
    static const uint32_t kGridMul = std::max<uint32_t>(1u, tune_u32("TQ_US_GRID_MUL", 16));
    share_grid = (uint32_t)std::min<uint64_t>(groups[kShare].n_chunks, (uint64_t)std::max(1, cus) * kGridMul);
    const size_t words = 2 * n_share + 16;
    rc = s->d_share_words.ensure(words * sizeof(uint32_t));
    if (rc == TQ_OK)
      rc = sc.share_stage.ensure((size_t)share_grid * TQD_US_GROUP * tqk_share_capl(groups[kShare].kpl) *
                                   sizeof(uint64_t));
    if (rc != TQ_OK) return rc;
    HIP_TRY(hipMemsetAsync(s->d_share_words.p, 0, words * sizeof(uint32_t), st));
  }

  uint32_t ashare_grid = 0;
  const size_t n_ashare = groups[kAShare].queries.size();
  if (n_ashare) {  // thr_val | list_count per query, then the task counter; staging lists of the persistent grid
    static const uint32_t kAGridMul = tune_u32("TQ_AS_GRID_MUL", 0);
    const uint32_t per_cu = kAGridMul ? kAGridMul : tqk_ashare_waves_per_cu();
    ashare_grid = (uint32_t)std::min<uint64_t>(groups[kAShare].n_chunks, (uint64_t)std::max(1, cus) * per_cu);
    const size_t words = 2 * n_ashare + 16;
    rc = s->d_ashare_words.ensure(words * sizeof(uint32_t));
    if (rc == TQ_OK)
      rc = sc.ashare_stage.ensure((size_t)ashare_grid * TQD_AS_GROUP * tqk_share_capl(groups[kAShare].kpl) *
                                    sizeof(uint64_t));
    if (rc != TQ_OK) return rc;
    HIP_TRY(hipMemsetAsync(s->d_ashare_words.p, 0, words * sizeof(uint32_t), st));
  }

  // ---- launch
  const uint8_t *ds = (const uint8_t *)dstage.p;
  if (s->opt.timing) HIP_TRY(hipEventRecord(s->ev_k0[slot], st));
  uint32_t tiles_total = 0, chunks_total = 0;
  // The scan kernels of the different launch groups are independent: all but the first run on
  // the segment's side stream, forked from and joined back into `st` with events, so that a
  // small group (e.g. the AND queries over sparse lists) fills the gaps of the big one instead
  // of adding its own ramp-up and tail.
  int n_active = 0;
  for (int gi = 0; gi < kGroups; ++gi) n_active += groups[gi].queries.empty() ? 0 : 1;
  const bool fork = n_active > 1;
  if (fork) {
    HIP_TRY(hipEventRecord(s->ev_fork, st));
    HIP_TRY(hipStreamWaitEvent(s->side_stream, s->ev_fork, 0));
  }
  const int launch_order[kGroups] = {kAndGeneral, kBool, kShare, kDense, 1, 2, kPhSweep, 0, kAShare};  // long serial chains first
  // the group that keeps the caller's stream: the batch's intersections
  const int main_group = n_ashare ? kAShare : 0;
  uint32_t kernel_mask = 0;
  for (int oi = 0; oi < kGroups; ++oi) {
    const int gi = launch_order[oi];
    Group &g = groups[gi];
    if (g.queries.empty()) continue;
    // the big dense-AND group keeps the caller's stream, the others go to the side stream
    hipStream_t gst = (fork && gi != main_group) ? s->side_stream : st;
    if (gi == kAShare) {
      TqkAShareParams ap{};
      ap.seg = s->dseg;
      ap.terms = s->d_terms;
      ap.queries = (const TqdQuery *)(ds + g.o_queries);
      ap.caches = (const float *)(ds + o_caches);
      ap.leads = (const TqdALead *)(ds + g.o_leads);
      ap.tasks = (const uint4 *)(ds + g.o_tasks);
      ap.sinks = (const TqkSinks *)(ds + g.o_sinks);
      ap.thr_slots = (uint32_t *)s->d_thr.p;
      ap.thr_val = (uint32_t *)s->d_ashare_words.p;
      ap.list_count = ap.thr_val + n_ashare;
      ap.table_base = (const uint8_t *)s->plan->share_table_base;
      ap.stage = (uint64_t *)sc.ashare_stage.p;
      ap.lists = (uint64_t *)((uint8_t *)sc.partials.p + part_off_bytes[gi]);
      ap.n_queries = (uint32_t)n_ashare;
      static const uint32_t kDebugA = tune_u32("TQ_DEBUG", 0);
      ap.debug = kDebugA;
      ap.bound_slack = co.bound_slack;
      tiles_total += g.total_tiles;
      chunks_total += g.n_chunks;
      kernel_mask |= TQ_KERNEL_ASHARE;
      // two launches: the warm-up tasks, then the rest (stream order = the barrier between them)
      const uint32_t bounds[3] = {0u, s->plan->a_warm_tasks, g.n_chunks};
      for (int ph = 0; ph < 2; ++ph) {
        ap.task_begin = bounds[ph];
        ap.n_tasks = bounds[ph + 1];
        if (ap.n_tasks <= ap.task_begin) continue;
        ap.task_counter = ap.thr_val + 2 * n_ashare + ph;
        ap.grid = std::min<uint32_t>(ashare_grid, ap.n_tasks - ap.task_begin);
        const hipError_t e = tqk_launch_ashare(ap, g.kpl, gst);
        if (e != hipSuccess) return fail(TQ_ERR_HIP, "shared-intersection launch: %s", hipGetErrorString(e));
      }
      continue;
    }
    if (gi == kShare) {
      kernel_mask |= TQ_KERNEL_USHARE;
      TqkShareParams sp{};
      sp.seg = s->dseg;
      sp.terms = s->d_terms;
      sp.queries = (const TqdQuery *)(ds + g.o_queries);
      sp.caches = (const float *)(ds + o_caches);
      sp.leads = (const TqdLead *)(ds + g.o_leads);
      sp.tasks = (const uint4 *)(ds + g.o_tasks);
      sp.sinks = (const TqkSinks *)(ds + g.o_sinks);
      sp.thr_slots = (uint32_t *)s->d_thr.p;
      sp.thr_val = (uint32_t *)s->d_share_words.p;
      sp.list_count = sp.thr_val + n_share;
      uint32_t *const counters = sp.thr_val + 2 * n_share;  // one task counter per launch
      sp.stage = (uint64_t *)sc.share_stage.p;
      sp.lists = (uint64_t *)((uint8_t *)sc.partials.p + part_off_bytes[gi]);
      sp.n_queries = (uint32_t)n_share;
      static const uint32_t kDebugS = tune_u32("TQ_DEBUG", 0);
      sp.debug = kDebugS;
      sp.bound_slack = co.bound_slack;
      tiles_total += g.total_tiles;
      chunks_total += g.n_chunks;
      // one launch per list position (stream order = the barrier between positions)
      static const uint32_t kPhases = tune_u32("TQ_US_PHASES", 0);
      for (uint32_t ph = 0; ph < TQD_US_MAX_TERMS; ++ph) {
        sp.task_begin = s->plan->share_phase_first[ph];
        sp.n_tasks = s->plan->share_phase_first[ph + 1];
        if (!kPhases) {  // (experiments) one launch, tasks still in position order
          if (ph) break;
          sp.n_tasks = g.n_chunks;
        }
        if (sp.n_tasks <= sp.task_begin) continue;
        sp.task_counter = counters + ph;
        sp.table_base = (const uint8_t *)s->plan->share_table_base;
        sp.grid = std::min<uint32_t>(share_grid, sp.n_tasks - sp.task_begin);
        const hipError_t e = tqk_launch_share(sp, g.kpl, gst);
        if (e != hipSuccess) return fail(TQ_ERR_HIP, "shared-union launch: %s", hipGetErrorString(e));
      }
      continue;
    }
    if (gi == kDense) {
      kernel_mask |= TQ_KERNEL_XUNION;
      TqkDenseParams dp{};
      dp.seg = s->dseg;
      dp.terms = s->d_terms;
      dp.rows = (const TqkDenseRow *)(ds + g.o_leads);
      dp.queries = (const TqkDenseQuery *)(ds + g.o_tasks);
      dp.cache = (const float *)(ds + o_caches) + (size_t)dense_cache * 256u;
      dp.sinks = (const TqkSinks *)(ds + g.o_sinks);
      dp.thr_slots = (uint32_t *)s->d_thr.p;
      dp.thr_val = (uint32_t *)s->d_share_words.p;
      dp.list_count = dp.thr_val + n_dense;
      dp.task_counter = dp.thr_val + 2 * n_dense;
      dp.stage = (uint64_t *)sc.share_stage.p;
      dp.lists = (uint64_t *)((uint8_t *)sc.partials.p + part_off_bytes[gi]);
      dp.n_rows = (uint32_t)s->plan->xrows.size();
      dp.n_bitmap_rows = s->plan->x_bitmap_rows;
      dp.n_queries = (uint32_t)n_dense;
      dp.max_terms = s->plan->x_max_terms;
      dp.n_tasks = g.n_chunks;
      dp.tiles_per_task = s->plan->x_tiles_per_task;
      dp.list_stride = s->plan->x_list_stride;
      dp.grid = s->plan->xgrid;
      static const uint32_t kDebugX = tune_u32("TQ_DEBUG", 0);
      dp.debug = kDebugX;
      tiles_total += g.total_tiles;
      chunks_total += g.n_chunks;
      const hipError_t e = tqk_launch_xunion(dp, g.kpl, gst);
      if (e != hipSuccess) return fail(TQ_ERR_HIP, "doc-major union launch: %s", hipGetErrorString(e));
      continue;
    }
    TqkScanParams p{};
    p.seg = s->dseg;
    if (!s->opt.use_dense) p.seg.docmat = nullptr;
    p.terms = s->d_terms;
    p.queries = (const TqdQuery *)(ds + g.o_queries);
    p.tile_starts = (const uint32_t *)(ds + g.o_tiles);
    p.caches = (const float *)(ds + o_caches);
    p.sinks = (const TqkSinks *)(ds + g.o_sinks);
    p.thr_slots = (uint32_t *)s->d_thr.p;
    p.n_queries = (uint32_t)g.queries.size();
    p.total_tiles = g.total_tiles;
    p.chunk_recs = (const uint4 *)(ds + g.o_chunks);
    p.n_chunks = g.n_chunks;
    p.exhaustive = (uint32_t)opt_exhaustive;
    p.use_dense = (uint32_t)s->opt.use_dense;
    p.all_dense = (gi == 0 || (gi == 2 && phrase_all_dense)) ? 1u : 0u;
    static const uint32_t kDebug = tune_u32("TQ_DEBUG", 0);
    p.debug = kDebug;
    p.or_windows = gi == kPhSweep ? 2u : ((or_windows_opt && gi != kBool) ? 1u : 0u);  // (2 = phrase sweep)
    p.boolean = gi == kBool ? 1u : 0u;
    p.small_k = g.max_k <= 16u ? 1u : 0u;
    p.bound_slack = co.bound_slack;
    p.max_terms = 0;
    for (const TqdQuery &dq : g.queries) p.max_terms = std::max(p.max_terms, dq.n_terms);
    tiles_total += g.total_tiles;
    chunks_total += g.n_chunks;
    kernel_mask |= gi == 0 ? TQ_KERNEL_AND_DENSE
                   : gi == kAndGeneral ? TQ_KERNEL_AND
                   : gi == kBool ? TQ_KERNEL_BOOL
                   : gi == kPhSweep ? TQ_KERNEL_PHRASE_SWEEP
                   : gi == 2 ? TQ_KERNEL_PHRASE
                   : (p.or_windows ? TQ_KERNEL_OR_WINDOWS : TQ_KERNEL_UNION);
    hipError_t e = hipSuccess;
    if (g.mode == TQ_MODE_AND)
      e = tqk_launch_and(p, g.kpl, s->opt.use_dpp != 0, gst);
    else if (g.mode == TQ_MODE_OR)
      e = tqk_launch_or(p, g.kpl, s->opt.use_dpp != 0, gst);
    else
      e = tqk_launch_phrase(p, g.kpl, s->opt.use_dpp != 0, gst);
    if (e != hipSuccess) return fail(TQ_ERR_HIP, "scan kernel launch: %s", hipGetErrorString(e));
  }
  if (fork) {
    HIP_TRY(hipEventRecord(s->ev_join, s->side_stream));
    HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));
  }
  if (s->opt.timing) HIP_TRY(hipEventRecord(s->ev_k1[slot], st));
  for (int gi = 0; gi < kGroups; ++gi) {
    Group &g = groups[gi];
    if (g.queries.empty()) continue;
    TqkMergeParams m{};
    m.queries = (const TqdQuery *)(ds + g.o_queries);
    m.partials = (const uint64_t *)((const uint8_t *)sc.partials.p + part_off_bytes[gi]);
    m.out_index = (const uint32_t *)(ds + g.o_outidx);
    m.out_scores = d_out_scores;
    m.out_docs = d_out_docs;
    m.out_counts = d_out_counts;
    m.n_queries = (uint32_t)g.queries.size();
    m.out_stride = out_stride;
    hipError_t e = gi == kAShare  ? tqk_launch_merge_lists(m, (const uint32_t *)s->d_ashare_words.p + n_ashare, g.kpl, st)
                   : gi == kShare ? tqk_launch_merge_lists(m, (const uint32_t *)s->d_share_words.p + n_share, g.kpl, st)
                   : gi == kDense ? tqk_launch_merge_lists(m, (const uint32_t *)s->d_share_words.p + n_dense, g.kpl, st)
                                  : tqk_launch_merge(m, g.kpl, st);
    if (e != hipSuccess) return fail(TQ_ERR_HIP, "merge kernel launch: %s", hipGetErrorString(e));
  }
  if (s->opt.timing) {
    HIP_TRY(hipEventRecord(s->ev_t1[slot], st));
    ++s->batches_timed;
  }
  HIP_TRY(hipEventRecord(s->ev_batch_done, st));
  HIP_TRY(hipEventRecord(sc.ev_last, st));
  sc.last_stream = st;
  sc.in_flight = true;
  if (kCopyStream) {
    HIP_TRY(hipEventRecord(s->ev_buf_free[bx], st));
    s->buf_used[bx] = true;
  }
  ++s->batches_enqueued;
  s->last_stream = st;
  s->batch_in_flight = true;
  if (trace) {
    const auto tr3 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    fprintf(stderr, "[tq] plan %ld us (pre-pass %ld us, queries %ld us, chunks %ld us), wait for the staging buffer %ld us, stage fill %ld us, enqueue %ld us, stage bytes %zu\n",
            us(tr0, tr1), us(tr0, tr0a), us(tr0a, tr0b), us(tr0b, tr1), us(tr1, tr1w), us(tr1w, tr2), us(tr2, tr3), stage);
  }
  s->stats.algorithmic_bytes = algo_bytes;
  s->stats.tiles = tiles_total;
  s->stats.chunks = chunks_total;
  s->stats.matches = 0;
  s->stats.kernel_ms = 0;
  s->stats.total_ms = 0;
  s->stats.host_plan_ms = 0;
  s->stats.kernel_mask = kernel_mask;
  s->stats.unique_bytes = unique_bytes;
  s->stats_pending = true;
  (void)total_parts;
  drain_on_error.armed = false;
  s->host_ms_sum += std::chrono::duration<double, std::milli>((std::chrono::steady_clock::now() - tr0) - (tr1w - tr1)).count();
  ++s->host_ms_n;
  return TQ_OK;
}

int resolve_opts(const tq_segment *s, const tq_search_opts *o, CallOpts &co) {
  co.exhaustive = s->opt.exhaustive != 0;
  uint32_t ppm = (uint32_t)s->opt.bound_slack_ppm;
  if (o) {
    if (o->exhaustive == 0 || o->exhaustive == 1)
      co.exhaustive = o->exhaustive != 0;
    else if (o->exhaustive != -1)
      return fail(TQ_ERR_INVALID, "tq_search_opts.exhaustive must be -1, 0 or 1");
    if (o->bound_slack_ppm != TQ_OPT_DEFAULT) ppm = o->bound_slack_ppm;
    if (ppm > 1000000000u) return fail(TQ_ERR_INVALID, "bound_slack_ppm above 1e9");
  }
  co.bound_slack = 1.0f + (float)ppm * 1e-6f;
  return TQ_OK;
}

int search_batch_host(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                      uint32_t out_stride, float *out_scores, uint32_t *out_docs,
                      uint32_t *out_counts, const CallOpts &co);

}  // namespace

extern "C" {

int tq_search_batch_device(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                           uint32_t out_stride, float *d_out_scores, uint32_t *d_out_docs,
                           uint32_t *d_out_counts, void *hip_stream) {
  return tq_search_batch_device_opts(s, queries, n_queries, out_stride, d_out_scores, d_out_docs,
                                     d_out_counts, nullptr, hip_stream);
}

int tq_search_batch_device_opts(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                                uint32_t out_stride, float *d_out_scores, uint32_t *d_out_docs,
                                uint32_t *d_out_counts, const tq_search_opts *opts,
                                void *hip_stream) {
  if (!s) return fail(TQ_ERR_INVALID, "tq_search_batch: null segment");
  TQ_SEGMENT_LOCK(s);
  CallOpts co;
  const int rc = resolve_opts(s, opts, co);
  if (rc != TQ_OK) return rc;
  return search_batch_impl(s, queries, n_queries, out_stride, d_out_scores, d_out_docs,
                           d_out_counts, hip_stream, co);
}

int tq_search_batch(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                    uint32_t out_stride, float *out_scores, uint32_t *out_docs,
                    uint32_t *out_counts) {
  return tq_search_batch_opts(s, queries, n_queries, out_stride, out_scores, out_docs, out_counts,
                              nullptr);
}

int tq_search_batch_opts(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                         uint32_t out_stride, float *out_scores, uint32_t *out_docs,
                         uint32_t *out_counts, const tq_search_opts *opts) {
  if (!s) return fail(TQ_ERR_INVALID, "tq_search_batch: null segment");
  TQ_SEGMENT_LOCK(s);
  CallOpts co;
  const int rc = resolve_opts(s, opts, co);
  if (rc != TQ_OK) return rc;
  return search_batch_host(s, queries, n_queries, out_stride, out_scores, out_docs, out_counts, co);
}

}  // extern "C"

namespace {
int search_batch_host(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                      uint32_t out_stride, float *out_scores, uint32_t *out_docs,
                      uint32_t *out_counts, const CallOpts &co) {
  if (!s || !out_scores || !out_docs || !out_counts)
    return fail(TQ_ERR_INVALID, "tq_search_batch: null argument");
  if (n_queries == 0) return TQ_OK;
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = (size_t)n_queries * out_stride;
  int rc = s->d_out_scores.ensure(n * sizeof(float));
  if (rc == TQ_OK) rc = s->d_out_docs.ensure(n * sizeof(uint32_t));
  if (rc == TQ_OK) rc = s->d_out_counts.ensure((size_t)n_queries * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  rc = search_batch_impl(s, queries, n_queries, out_stride, (float *)s->d_out_scores.p,
                         (uint32_t *)s->d_out_docs.p, (uint32_t *)s->d_out_counts.p, nullptr, co);
  if (rc != TQ_OK) return rc;
  HIP_TRY(hipMemcpyAsync(out_scores, s->d_out_scores.p, n * sizeof(float), hipMemcpyDeviceToHost,
                         s->stream));
  HIP_TRY(hipMemcpyAsync(out_docs, s->d_out_docs.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                         s->stream));
  HIP_TRY(hipMemcpyAsync(out_counts, s->d_out_counts.p, (size_t)n_queries * sizeof(uint32_t),
                         hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}
}  // namespace

extern "C" {

int tq_segment_set_alive_bitset(tq_segment *s, const uint8_t *bytes, size_t len) {
  if (!s) return fail(TQ_ERR_INVALID, "tq_segment_set_alive_bitset: null segment");
  TQ_SEGMENT_LOCK(s);
  HIP_TRY(hipSetDevice(s->device));
  {
    const int wrc = wait_segment_idle(s);  // batches may run on a caller's stream
    if (wrc != TQ_OK) return wrc;
  }
  if (s->d_alive) (void)hipFree(s->d_alive);
  s->d_alive = nullptr;
  s->bytes_alive = 0;
  s->dseg.alive = nullptr;
  if (!bytes) return TQ_OK;  // no deletes
  // BitSet::serialize (common/src/bitset.rs:215-223): u32 LE max_value, then 64-bit tiny sets
  if (len < 4) return fail(TQ_ERR_FORMAT, "alive bitset shorter than its header");
  const uint32_t max_value = rd32(bytes);
  const size_t need = ((size_t)s->max_doc + 63) / 64 * 8;
  if (max_value != s->max_doc || len - 4 < need || (len - 4) % 8 != 0)
    return fail(TQ_ERR_FORMAT, "alive bitset for %u docs / %zu bytes does not fit max_doc %u",
                max_value, len - 4, s->max_doc);
  HIP_TRY(hipMalloc((void **)&s->d_alive, len - 4 + PAD));
  HIP_TRY(hipMemset(s->d_alive + (len - 4), 0, PAD));
  HIP_TRY(hipMemcpy(s->d_alive, bytes + 4, len - 4, hipMemcpyHostToDevice));
  s->bytes_alive = len - 4;
  s->dseg.alive = s->d_alive;
  return TQ_OK;
}

int tq_last_batch_match_counts(tq_segment *s, uint32_t *out, uint32_t n) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_last_batch_match_counts: null argument");
  TQ_SEGMENT_LOCK(s);
  if (n > s->last_batch_queries) return fail(TQ_ERR_INVALID, "the last batch had %u queries", s->last_batch_queries);
  HIP_TRY(hipSetDevice(s->device));
  {
    const int wrc = wait_segment_idle(s);
    if (wrc != TQ_OK) return wrc;
  }
  HIP_TRY(hipMemcpy(out, s->d_qmatches.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return TQ_OK;
}

int tq_count_batch(tq_segment *s, const tq_query *queries, uint32_t n_queries,
                   uint32_t *out_counts) {
  if (!s || (!queries && n_queries) || !out_counts)
    return fail(TQ_ERR_INVALID, "tq_count_batch: null argument");
  TQ_SEGMENT_LOCK(s);
  if (n_queries == 0) return TQ_OK;
  // every match has to be visited: exhaustive scan, smallest top-k
  std::vector<tq_query> qs(queries, queries + n_queries);
  for (tq_query &q : qs) q.k = 1;
  std::vector<float> sc(n_queries);
  std::vector<uint32_t> dc(n_queries), ct(n_queries);
  CallOpts co;
  int rc = resolve_opts(s, nullptr, co);
  if (rc != TQ_OK) return rc;
  co.exhaustive = true;  // per call: the segment's options are not touched
  rc = search_batch_host(s, qs.data(), n_queries, 1, sc.data(), dc.data(), ct.data(), co);
  if (rc != TQ_OK) return rc;
  return tq_last_batch_match_counts(s, out_counts, n_queries);
}

int tq_last_batch_stats(tq_segment *s, tq_batch_stats *out) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_last_batch_stats: null argument");
  TQ_SEGMENT_LOCK(s);
  HIP_TRY(hipSetDevice(s->device));
  if (s->stats_pending) {
    {
      const int wrc = wait_segment_idle(s);  // batches may run on a caller's stream
      if (wrc != TQ_OK) return wrc;
    }
    unsigned long long m = 0;
    HIP_TRY(hipMemcpy(&m, s->d_match_counter, sizeof m, hipMemcpyDeviceToHost));
    s->stats.matches = m;
    s->stats.algorithmic_bytes += m;  // 1 fieldnorm byte per scored doc
    if (s->opt.timing) {
      uint64_t first = s->batches_reported;
      if (s->batches_timed - first > (uint64_t)tq_segment::kTimingRing)
        first = s->batches_timed - tq_segment::kTimingRing;
      double k_sum = 0, t_sum = 0;
      uint32_t n = 0;
      for (uint64_t b = first; b < s->batches_timed; ++b) {
        const int i = (int)(b % tq_segment::kTimingRing);
        float km = 0, tm = 0;
        if (hipEventElapsedTime(&km, s->ev_k0[i], s->ev_k1[i]) == hipSuccess &&
            hipEventElapsedTime(&tm, s->ev_t0[i], s->ev_t1[i]) == hipSuccess) {
          k_sum += km;
          t_sum += tm;
          ++n;
        }
      }
      s->batches_reported = s->batches_timed;
      if (n) {
        s->stats.kernel_ms = (float)(k_sum / n);
        s->stats.total_ms = (float)(t_sum / n);
      }
      s->stats.batches_averaged = n;
    }
    s->stats.host_plan_ms = s->host_ms_n ? (float)(s->host_ms_sum / s->host_ms_n) : 0.0f;
    s->host_ms_sum = 0;
    s->host_ms_n = 0;
    s->stats_pending = false;
  }
  *out = s->stats;
  return TQ_OK;
}

int tq_segment_reserve_columns(tq_segment *s, const uint64_t *postings_offs, uint32_t n) {
  if (!s || (!postings_offs && n)) return fail(TQ_ERR_INVALID, "tq_segment_reserve_columns: null argument");
  TQ_SEGMENT_LOCK(s);
  if (!s->terms.empty())
    return fail(TQ_ERR_INVALID, "tq_segment_reserve_columns: call it before the first tq_term_prepare");
  s->reserved_cols.clear();
  for (uint32_t i = 0; i < n && i < TQD_MAT_SLOTS; ++i) s->reserved_cols[postings_offs[i]] = true;
  s->cols_reserved = true;
  return TQ_OK;
}

int tq_segment_get_stats(tq_segment *s, tq_segment_stats *out) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_segment_get_stats: null argument");
  TQ_SEGMENT_LOCK(s);
  tq_segment_stats r{};
  r.index_bytes = s->idx_len;
  r.positions_bytes = s->pos_len;
  r.fieldnorm_bytes = s->d_fn ? s->max_doc : 0;
  r.alive_bytes = s->bytes_alive;
  r.term_table_bytes = s->bytes_term_tables + s->d_terms_cap * sizeof(TqdTerm);
  r.bitmap_bytes = s->bytes_bitmaps;
  r.docmat_bytes = s->bytes_docmat;
  r.posdir_bytes = s->bytes_posdir;
  r.scratch_bytes = s->d_stage.cap + s->d_stage_alt.cap + s->d_out_scores.cap + s->d_out_docs.cap +
                    s->d_out_counts.cap + s->d_misc.cap + s->d_thr.cap + s->d_qmatches.cap + s->d_share_words.cap +
                    s->d_ashare_words.cap;
  {
    std::lock_guard<std::mutex> lk(s->dscratch->m);
    r.device_scratch_bytes = s->dscratch->partials.cap + s->dscratch->share_stage.cap + s->dscratch->ashare_stage.cap;
  }
  r.n_terms = (uint32_t)s->terms.size();
  r.n_dense_lists = s->n_dense_lists;
  r.n_docmat_columns = s->n_mat_slots;
  r.dense_budget_bytes = s->dense_budget();
  *out = r;
  return TQ_OK;
}

int tq_set_option(tq_segment *s, const char *name, int64_t value) {
  if (!s || !name) return fail(TQ_ERR_INVALID, "tq_set_option: null argument");
  TQ_SEGMENT_LOCK(s);
  if (!strcmp(name, "exhaustive"))
    s->opt.exhaustive = value != 0;
  else if (!strcmp(name, "timing"))
    s->opt.timing = value != 0;
  else if (!strcmp(name, "use_dpp"))
    s->opt.use_dpp = value != 0;
  else if (!strcmp(name, "use_dense"))
    s->opt.use_dense = value != 0;
  else if (!strcmp(name, "or_windows"))
    s->opt.or_windows = value < 0 ? -1 : (value != 0);
  else if (!strcmp(name, "dense_ratio") && value >= 1)  // affects terms prepared afterwards
    s->opt.dense_ratio = (int)value;
  else if (!strcmp(name, "dense_budget_x") && value >= 0)
    s->opt.dense_budget_x = (int)value;
  else if (!strcmp(name, "bound_slack_ppm") && value >= 0 && value <= 1000000000)
    s->opt.bound_slack_ppm = (int)value;
  else if (!strcmp(name, "dense"))  // affects terms prepared afterwards
    s->opt.dense = value != 0;
  else if (!strcmp(name, "docmat"))  // affects terms prepared afterwards
    s->opt.docmat = value != 0;
  else if (!strcmp(name, "docsig"))  // affects terms prepared afterwards
    s->opt.docsig = value != 0;
  else if (!strcmp(name, "device_prepare"))  // affects terms prepared afterwards
    s->opt.device_prepare = value != 0;
  else if (!strcmp(name, "xunion_ratio") && value >= 0 && value <= 0x7FFFFFFF)
    s->opt.xunion_ratio = (int)value;
  else if (!strcmp(name, "xunion_min_queries") && value >= 1 && value <= 0x7FFFFFFF)
    s->opt.xunion_min_queries = (int)value;
  else if (!strcmp(name, "submit_window_us") && value >= 0 && value <= 1000000)
    s->opt.submit_window_us = (int)value;
  else
    return fail(TQ_ERR_INVALID, "unknown option '%s'", name);
  return TQ_OK;
}

int tq_decode_postings(tq_segment *s, tq_term_handle term, uint32_t *docs, uint32_t *tfs) {
  if (!s || !docs || !tfs) return fail(TQ_ERR_INVALID, "tq_decode_postings: null argument");
  TQ_SEGMENT_LOCK(s);
  if (term >= s->terms.size()) return fail(TQ_ERR_INVALID, "unknown term handle %u", term);
  HIP_TRY(hipSetDevice(s->device));
  int rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  const TermHost &t = s->terms[term];
  const size_t bytes = (size_t)t.doc_freq * sizeof(uint32_t);
  rc = s->d_misc.ensure(2 * bytes + 64);
  if (rc != TQ_OK) return rc;
  uint32_t *dd = (uint32_t *)s->d_misc.p, *dt = dd + t.doc_freq;
  hipError_t e = tqk_launch_decode_list(s->dseg, s->d_terms, term, t.n_blocks, dd, dt,
                                        s->opt.use_dpp != 0, s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(docs, dd, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(tfs, dt, bytes, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}

int tq_decode_position_deltas(tq_segment *s, tq_term_handle term, uint32_t *out, uint64_t cap,
                              uint64_t *n_out) {
  if (!s || !n_out) return fail(TQ_ERR_INVALID, "tq_decode_position_deltas: null argument");
  TQ_SEGMENT_LOCK(s);
  if (term >= s->terms.size()) return fail(TQ_ERR_INVALID, "unknown term handle %u", term);
  const TermHost &t = s->terms[term];
  if (t.positions_len == 0) return fail(TQ_ERR_UNSUPPORTED, "term has no positions on the device");
  *n_out = t.n_positions;
  const uint64_t n = std::min<uint64_t>(cap, t.n_positions);
  if (n == 0 || !out) return TQ_OK;
  HIP_TRY(hipSetDevice(s->device));
  int rc = sync_terms(s, s->stream);
  if (rc != TQ_OK) return rc;
  rc = s->d_misc.ensure(n * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  hipError_t e = tqk_launch_decode_positions(s->dseg, s->d_terms, term, (uint32_t *)s->d_misc.p, n,
                                             s->stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "decode launch: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(out, s->d_misc.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return TQ_OK;
}

}  // extern "C"

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
  bool done = false;
};
struct SubmitQueue {
  std::mutex m;
  std::condition_variable cv;
  std::condition_variable cv_arrive;  // a query was submitted (the leader's arrival window)
  std::deque<tq_ticket *> pending;
  bool leader_active = false;
  size_t last_batch = 0;  // queries the previous batch carried
  tq_submit_stats stats{};
  // the leader's scratch
  std::vector<tq_query> qs;
  std::vector<float> sc;
  std::vector<uint32_t> dc, ct;
};
void tq_free_submit_queue(SubmitQueue *q) { delete q; }

namespace {
constexpr size_t kSubmitMaxBatch = 16384;
std::mutex g_submit_create_m;

SubmitQueue *submit_queue(tq_segment *s) {
  std::lock_guard<std::mutex> lk(g_submit_create_m);
  if (!s->submit) s->submit = new SubmitQueue();
  return s->submit;
}

// one launch for the tickets of `batch` (same options); rows go to the callers' buffers
void run_ticket_batch(SubmitQueue &Q, tq_segment *s, std::vector<tq_ticket *> &batch) {
  const uint32_t n = (uint32_t)batch.size();
  uint32_t stride = 1;
  Q.qs.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    Q.qs[i] = batch[i]->q;
    stride = std::max(stride, batch[i]->q.k);
  }
  Q.sc.resize((size_t)n * stride);
  Q.dc.resize((size_t)n * stride);
  Q.ct.resize(n);
  int rc;
  {
    TQ_SEGMENT_LOCK(s);
    rc = search_batch_host(s, Q.qs.data(), n, stride, Q.sc.data(), Q.dc.data(), Q.ct.data(), batch[0]->co);
  }
  if (rc == TQ_OK) {
    for (uint32_t i = 0; i < n; ++i) {
      tq_ticket *t = batch[i];
      const uint32_t k = t->q.k;
      memcpy(t->out_scores, Q.sc.data() + (size_t)i * stride, k * sizeof(float));
      memcpy(t->out_docs, Q.dc.data() + (size_t)i * stride, k * sizeof(uint32_t));
      *t->out_count = Q.ct[i];
      t->rc = TQ_OK;
    }
    return;
  }
  if (n == 1) {
    batch[0]->rc = rc;
    batch[0]->err = g_last_error;
    return;
  }
  // one query the device does not take (an unsupported shape, a bad handle) must not fail its
  // neighbours: the batch is run again query by query, every caller gets its own verdict
  for (uint32_t i = 0; i < n; ++i) {
    std::vector<tq_ticket *> one{batch[i]};
    run_ticket_batch(Q, s, one);
  }
}

int ticket_wait(tq_ticket *t) {
  tq_segment *s = t->seg;
  SubmitQueue &Q = *s->submit;
  std::unique_lock<std::mutex> lk(Q.m);
  std::vector<tq_ticket *> batch;
  while (!t->done) {
    if (Q.leader_active || Q.pending.empty()) {
      Q.cv.wait(lk);
      continue;
    }
    // lead one batch: everything pending that runs under the first ticket's options.  Callers of
    // the batch that just finished are on their way back with their next query: the leader gives them
    // up to submit_window_us to arrive (until as many are pending as the last batch carried) — without
    // it the first caller back leads a batch of one and everybody else waits a whole launch longer
    Q.leader_active = true;
    if (Q.pending.size() < Q.last_batch && s->opt.submit_window_us > 0) {
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(s->opt.submit_window_us);
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
    lk.unlock();
    run_ticket_batch(Q, s, batch);
    lk.lock();
    ++Q.stats.batches;
    Q.stats.queries += batch.size();
    Q.stats.max_batch = std::max<uint64_t>(Q.stats.max_batch, batch.size());
    for (tq_ticket *b : batch) b->done = true;
    Q.last_batch = batch.size();
    Q.leader_active = false;
    Q.cv.notify_all();
  }
  const int rc = t->rc;
  if (rc != TQ_OK) g_last_error = t->err;  // (this thread's slot)
  return rc;
}
}  // namespace

extern "C" {

int tq_submit(tq_segment *s, const tq_query *q, const tq_search_opts *opts, float *out_scores,
              uint32_t *out_docs, uint32_t *out_count, tq_ticket **out) {
  if (!s || !q || !out_scores || !out_docs || !out_count || !out)
    return fail(TQ_ERR_INVALID, "tq_submit: null argument");
  // what can be judged without the segment's state is judged here: a bad query never joins a batch
  if (q->n_terms == 0 || q->n_terms > TQ_MAX_TERMS)
    return fail(TQ_ERR_INVALID, "tq_submit: n_terms %u not in 1..%u", q->n_terms, TQ_MAX_TERMS);
  if (q->k == 0 || q->k > TQ_MAX_K) return fail(TQ_ERR_INVALID, "tq_submit: k %u not in 1..%u", q->k, TQ_MAX_K);
  if (!q->terms || !q->weights || !q->tf_cache) return fail(TQ_ERR_INVALID, "tq_submit: null terms/weights/tf_cache");
  CallOpts co;
  {
    TQ_SEGMENT_LOCK(s);
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
  }
  Q->cv_arrive.notify_one();  // (a leader may be holding its batch open for this query)
  *out = t;
  return TQ_OK;
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
  const int rc = tq_submit(s, q, opts, out_scores, out_docs, out_count, &t);
  if (rc != TQ_OK) return rc;
  return tq_wait(t);
}

int tq_get_submit_stats(tq_segment *s, tq_submit_stats *out, int reset) {
  if (!s || !out) return fail(TQ_ERR_INVALID, "tq_get_submit_stats: null argument");
  SubmitQueue *Q = submit_queue(s);
  std::lock_guard<std::mutex> lk(Q->m);
  *out = Q->stats;
  if (reset) Q->stats = tq_submit_stats{};
  return TQ_OK;
}

}  // extern "C"

extern "C" {
// ---- cross-segment merge (merge_top_k)
int tq_merge_topk(const float *scores, const uint32_t *docs, const uint32_t *counts,
                  uint32_t n_segments, uint32_t n_queries, uint32_t stride, uint32_t offset,
                  uint32_t limit, float *out_scores, uint32_t *out_segment_ords,
                  uint32_t *out_docs, uint32_t *out_counts) {
  if (!scores || !docs || !counts || !out_scores || !out_segment_ords || !out_docs || !out_counts)
    return fail(TQ_ERR_INVALID, "tq_merge_topk: null argument");
  struct H {
    float s;
    uint32_t o, d;
  };
  std::vector<H> all;
  for (uint32_t q = 0; q < n_queries; ++q) {
    all.clear();
    for (uint32_t sg = 0; sg < n_segments; ++sg) {
      const uint32_t c = std::min(counts[(size_t)sg * n_queries + q], stride);
      const size_t b = ((size_t)sg * n_queries + q) * stride;
      for (uint32_t i = 0; i < c; ++i) all.push_back({scores[b + i], sg, docs[b + i]});
    }
    // top_score_collector.rs:590-600: sort key desc, then DocAddress asc.  Scores are compared
    // through the order-preserving u32 transform the device keys use (a strict weak order even
    // for NaN inputs, which partial_cmp().unwrap_or(Equal) in the reference is not)
    auto sortable = [](float x) {
      uint32_t u;
      memcpy(&u, &x, 4);
      if ((u & 0x7FFFFFFFu) == 0u) u = 0u;  // -0.0 == +0.0
      return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
    };
    std::sort(all.begin(), all.end(), [&](const H &a, const H &b) {
      const uint32_t ka = sortable(a.s), kb = sortable(b.s);
      if (ka != kb) return ka > kb;
      if (a.o != b.o) return a.o < b.o;
      return a.d < b.d;
    });
    uint32_t w = 0;
    for (size_t i = offset; i < all.size() && w < limit; ++i, ++w) {
      out_scores[(size_t)q * limit + w] = all[i].s;
      out_segment_ords[(size_t)q * limit + w] = all[i].o;
      out_docs[(size_t)q * limit + w] = all[i].d;
    }
    out_counts[q] = w;
    for (; w < limit; ++w) {
      out_scores[(size_t)q * limit + w] = 0.0f;
      out_segment_ords[(size_t)q * limit + w] = 0xFFFFFFFFu;
      out_docs[(size_t)q * limit + w] = TQ_TERMINATED;
    }
  }
  return TQ_OK;
}

int tq_merge_topk_device(tq_ctx *ctx, int device, const float *d_scores, const uint32_t *d_docs,
                         const uint32_t *d_counts, const uint32_t *segment_ords,
                         uint32_t n_segments, uint32_t n_queries, uint32_t stride,
                         uint32_t offset, uint32_t limit, float *d_out_scores,
                         uint32_t *d_out_segment_ords, uint32_t *d_out_docs,
                         uint32_t *d_out_counts, void *hip_stream) {
  if (!ctx || !d_scores || !d_docs || !d_counts || !d_out_scores || !d_out_segment_ords ||
      !d_out_docs || !d_out_counts)
    return fail(TQ_ERR_INVALID, "tq_merge_topk_device: null argument");
  if (limit == 0) return fail(TQ_ERR_INVALID, "tq_merge_topk_device: limit 0");
  HIP_TRY(hipSetDevice(device));
  TqkSegMergeParams p{};
  p.scores = d_scores;
  p.docs = d_docs;
  p.counts = d_counts;
  p.segment_ords = segment_ords;
  p.out_scores = d_out_scores;
  p.out_segment_ords = d_out_segment_ords;
  p.out_docs = d_out_docs;
  p.out_counts = d_out_counts;
  p.n_segments = n_segments;
  p.n_queries = n_queries;
  p.stride = stride;
  p.offset = offset;
  p.limit = limit;
  hipError_t e = tqk_launch_merge_segments(p, (hipStream_t)hip_stream);
  if (e != hipSuccess) return fail(TQ_ERR_HIP, "merge launch: %s", hipGetErrorString(e));
  return TQ_OK;
}

}  // extern "C"
