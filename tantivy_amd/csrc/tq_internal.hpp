// tq_internal.hpp — declarations shared by the translation units of the C ABI library
// (include/tantivy_amd.h): the segment and context objects, device buffers, the host planner's
// scratch and its launch groups, the planner thread pool, and the internal entry points of
//   tq_api.cpp          the ABI's object lifecycle, options, statistics, codec access, merges
//   tq_terms.cpp        tq_term_prepare: skip-list unrolling, dense side tables, doc matrix
//   tq_plan_chunks.cpp  tiles -> chunks -> launch order of the per-query kernels
//   tq_plan_share.cpp   the term-major launches: shared unions (leads / tasks), shared intersections
//   tq_plan_misc.cpp    the doc-major union plan, boolean query layout
//   tq_search.cpp       one batch: validate, plan, stage, launch (tq_search_batch*)
//   tq_submit.cpp       tq_submit / tq_wait / tq_search_one: coalescing of concurrent single queries
// Everything here is internal: nothing outside tantivy_amd/csrc (and tools/planbench, which tests the
// planner without a GPU) includes it.
#pragma once
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
#include <unordered_set>
#include <thread>
#include <vector>

#include "../../include/tantivy_amd.h"
#include "tq_device.h"
#include "tq_launch.h"
#include "tq_prepare.h"

namespace tqi {

extern thread_local std::string g_last_error;
int fail(int code, const char *fmt, ...);
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
  // (the old buffer goes first — a grown buffer next to its predecessor would double the peak — and the
  // capacity with it: a failed allocation leaves {null, 0}, never a stale capacity over a null pointer)
  int ensure(size_t n) {
    if (n <= cap) return TQ_OK;
    const size_t ncap = std::max(n, cap * 2);
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    void *np = nullptr;
    HIP_TRY(hipMalloc(&np, ncap));
    p = np;
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
    const size_t ncap = std::max(n, cap * 2);
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    void *np = nullptr;
    HIP_TRY(hipHostMalloc(&np, ncap, hipHostMallocDefault));
    p = np;
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
  const TqdTerm *d_self = nullptr;  // ... and a copy of the term's record as of its upload (rec / coarse / tail /
                                    // payload fields: what the kernels that build its side tables read)
  void *dense_blob = nullptr;  // bitmap + rank directory of a dense list
  void *posdir_blob = nullptr; // position directory of a dense list with positions
  void *tf8_blob = nullptr;    // term freqs of a dense list as bytes (posting index -> min(tf, 255))
  void *pos_blob = nullptr;    // device-side prepare: positions tables (sized after the walk)
  void *probe_dense_blob = nullptr, *probe_tf8_blob = nullptr;  // a list below "dense_ratio" that boolean queries
                               // probe in the shared launch: bitmap + rank directory and tf bytes built on first
                               // use ("probe_budget_x"); the other kernels do not see them
  void *probe_posdir_blob = nullptr;  // ... and its position directory (TqdTerm::pos_dir layout), built the first time
                                      // a phrase INSIDE a boolean query names the list (tq_tree.hip)
  int32_t probe_slot = -1;            // the slot of the segment's probe pool that holds them (tq_terms.cpp), or -1
  // a list below "dense_ratio": its range directory (rdir_lookup, tq_common.hpp: one u32 per posting in posting order +
  // a directory of posting counts per 2^rdir_shift docs; 6-8 bytes per posting), built when the term is prepared, while
  // such tables stay within "rdir_budget_x": the shared intersection launch asks it "is d in the list, with which tf" —
  // what a max_doc / 4-byte bitmap + rank directory + tf bytes from the probe pool answered before
  void *rdir_blob = nullptr;          // the directory (256-byte aligned: the shift rides in the low bits of its offset)
  void *rdir_ent = nullptr;           // the entries, right behind it
  uint32_t rdir_shift = 0;
  uint8_t *probe_own_dir = nullptr;   // a list too long for a slot: room for its position directory next to its own tables
  void *rmax_blob = nullptr;   // range maxima of a list with a bitmap (its own or the probe tables'): one byte per
                               // TQD_RM_SHIFT docs, tq_ashare.hip's bound on non-leader lists
  uint32_t rmax_list = 255;    // ... the largest of them
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
inline bool read_vint(const uint8_t *d, size_t len, size_t &at, uint64_t &out) {
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
inline bool read_vint32_block(const uint8_t *d, size_t len, size_t &at, uint32_t &out) {
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
  int rdir_budget_x = 4;    // range directories of the lists below "dense_ratio" (6-8 B per posting): at most this multiple of the segment
  int probe_budget_x = 16;  // bitmaps + tf bytes built on demand for the lists boolean queries probe: at most this multiple of the segment
  int count_bitmap_ratio = 128;  // Count: bitmap words instead of a scan if the driving clause holds >= max_doc / ratio postings per list
  int ashare_min_batch = 16;    // intersections: the shared launch needs this many qualifying queries in the batch (512 until round 6)
  // tq_submit / tq_search_one: how long the leader of a batch waits for the callers of the previous
  // batch to come back with their next query (0 = launch with whatever is pending)
  int submit_window_us = 100;
  int record_query_kernels = 0;  // tq_last_batch_query_kernels: remember which scan-kernel family ran every query
  int debug = -1;  // >= 0: overrides TQ_DEBUG for this segment's launches (work counters / ablations: diagnosis only)
};


inline uint32_t tune_u32(const char *name, uint32_t dflt) {
  const char *v = getenv(name);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

}  // namespace tqi
using namespace tqi;  // (the objects below are the ABI's opaque types: global)

// The big per-batch buffers — partial / result lists, the staging lists of the two term-major launches —
// exist once per DEVICE, not once per segment: a device runs one batch at a time anyway (the kernels fill
// it), and 100 segments on a GPU must not mean 100 copies (8 segments held 13.9 GB of scratch in round 3).
// A batch takes the lock when it sizes the buffers and keeps it until the event behind its last kernel is
// recorded; a batch on another stream than the previous user's first waits for that event (stream side).
struct DeviceScratch {
  std::mutex m;
  DevBuf partials, share_stage, ashare_stage, bshare_stage;
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
      kv.second->bshare_stage.release();
      delete kv.second;
    }
  }
};

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
  float *d_local_cache = nullptr;  // Bm25Weight.cache under the segment's OWN average fieldnorm (256 floats; null:
                                   // the header holds no usable token count): what the range maxima are built under
  uint64_t *d_docmat = nullptr;  // doc-major matrix of the dense lists (TqdSegment::docmat)
  uint64_t *d_doccls = nullptr;  // tf classes of the first TQD_CLS_SLOTS column lists (TqdSegment::doccls)
  uint32_t n_mat_slots = 0;
  TqdSegment dseg{};
  std::vector<TermHost> terms;
  std::vector<std::pair<void *, size_t>> rdir_chunks;  // the range directories' chunks (rdir_alloc, tq_terms.cpp)
  uint8_t *rdir_chunk_cur = nullptr;
  size_t rdir_chunk_left = 0;
  bool rdir_span_ok = false;       // ... all within the 32 GB the shared launches' table offsets reach (tq_search.cpp)
  std::vector<void *> term_slabs;  // the terms' table blobs are carved out of 4 MB slabs (term_alloc)
  uint8_t *term_slab_cur = nullptr;
  size_t term_slab_left = 0;
  std::vector<TqdTerm> h_dterms;
  TqdTerm *d_terms = nullptr;
  size_t d_terms_cap = 0;
  bool d_terms_dirty = false;
  size_t d_terms_dirty_from = ~(size_t)0;  // the lowest term record changed since the last sync
  size_t d_terms_synced = 0;               // records the device table holds (and batches in flight may read)
  size_t dense_bytes_total = 0;
  // The side tables of the dense lists (bitmaps + rank directories, byte-wide tfs, position directories,
  // plain lists) live in ONE device allocation of dense_budget() bytes, made with the first of them:
  // the term-major launches address them as 32-bit offsets (8-byte units) from its base.  (With one
  // hipMalloc per table the allocator now and then returned addresses more than 32 GB apart and the
  // launches silently fell back to the per-query kernels.)  What does not fit falls back to hipMalloc.
  uint8_t *dense_arena = nullptr;
  size_t dense_arena_cap = 0, dense_arena_used = 0;
  std::vector<void *> dense_extra;  // tables allocated outside the arena
  // Round 6: the arena is a RESERVED ADDRESS RANGE (hipMemAddressReserve, 24 GB of addresses — no memory), mapped
  // chunk by chunk as tables are added (hipMemCreate / hipMemMap): whatever the process has allocated and freed before,
  // every table of the segment lies within the 32 GB the shared launches' and the tree kernel's 32-bit table offsets
  // reach.  (A soak that opened and closed 230 indexes in one process got an overflow allocation more than 32 GB from
  // its arena: flat queries fell back to the per-query kernels, a nested query was refused.)  dense_arena_cap = the
  // reservation, dense_arena_mapped = what is backed by memory.  If the driver refuses any step: the old arena.
  bool dense_arena_vmm = false;
  void *dense_arena_va = nullptr;  // the reservation as the driver returned it (the arena starts at its first 32 MB multiple)
  size_t dense_arena_mapped = 0, dense_arena_first = 0;
  std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> dense_arena_chunks;
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
  DevBuf d_count_queries, d_count_out, d_count_bits, d_count_wgs;  // Count collector over bitmaps (tq_count.hip)
  DevBuf d_ashare_words, d_bshare_words;  // shared-intersection launches (run next to the shared-union one)
  DeviceScratch *dscratch = nullptr;  // partial / result lists and staging lists: the device's (tq_ctx)
  // the shared-union launch addresses bitmaps / byte-wide tfs as 32-bit offsets (8-byte units) from
  // the lowest such table: usable while all of them lie within 32 GB of device addresses
  size_t share_span_terms = 0;  // number of terms the span was computed over
  uint64_t share_table_lo = 0;
  bool share_span_ok = true;
  uint32_t last_batch_queries = 0;
  std::vector<uint32_t> last_query_kernel;  // option "record_query_kernels": TQ_KERNEL_* of every query of the last batch
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
  size_t probe_budget() const { return (size_t)opt.probe_budget_x * (idx_len + pos_len + max_doc); }
  size_t rdir_budget() const { return (size_t)opt.rdir_budget_x * (idx_len + pos_len + max_doc); }
  size_t rdir_bytes_total = 0;
  // tq_term_prepare_batch: the blobs of a batch's new terms on their way up (two buffers, an event each), and which
  // of the two events the next batch has to wait for (-1: none)
  PinnedBuf h_prep_stage[2];
  hipEvent_t ev_prep[2] = {nullptr, nullptr}, ev_prep_order = nullptr;
  bool prep_used[2] = {false, false};
  uint32_t prep_calls = 0;
  int prep_pending = -1;
  // the largest tf/(tf + norm) of every list whose range directory a tq_term_prepare_batch call built: written by the build
  // launch, copied into the call's pinned buffer behind the blobs; TermHost::rmax_list gets it once the call's event has
  // completed (prep_apply_lmax: no wait — until then the list's weight bounds it)
  std::vector<uint32_t> prep_lmax_handles[2];
  const uint32_t *prep_lmax_host[2] = {nullptr, nullptr};
  size_t probe_bytes_total = 0;
  bool probe_full = false;  // (kept for tq_set_option; the pool below evicts instead of latching)
  // Probe pool (round 6): the private tables of lists below "dense_ratio" live in equal SLOTS — [bitmap + rank
  // directory | tf bytes | position directory | range maxima] — "probe_budget_x" worth of them (at least one query's:
  // TQ_MAX_TERMS).  A list that needs tables when every slot is taken gets the slot of the list that was used longest
  // ago (by batch); slots touched by the batch being planned are never taken — a batch that names more such lists
  // than the budget holds grows the pool for good.  Before, the budget latched ("probe_full") and every later
  // query that named a new sparse list fell back or — nested boolean queries — failed.
  struct ProbeSlot {
    uint8_t *base = nullptr;
    uint32_t owner = 0xFFFFFFFFu;  // term handle, or none
    uint64_t last_batch = 0;
  };
  std::vector<ProbeSlot> probe_slots;
  size_t probe_slot_bytes = 0, probe_bm_bytes = 0, probe_tf_cap = 0, probe_dir_cap = 0, probe_rm_bytes = 0;
  uint64_t probe_batch = 1;        // sequence number of the batch being planned
  uint64_t probe_evictions = 0;
  uint64_t probe_replaced_batch = 0;  // ... and how many slots such candidates took over in it
  uint32_t probe_replaced_n = 0;
  uint64_t probe_no_room_batch = 0;  // the batch in which a shared-launch candidate found no slot to take
  bool probe_waited = false;       // this batch already waited for the batches in flight before reusing a slot
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
struct SubmitQueue *tq_new_submit_queue();
#define TQ_SEGMENT_LOCK(seg) std::lock_guard<std::recursive_mutex> tq_exec_lock_((seg)->exec_m)

namespace tqi {

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
  std::vector<TqdTreeQuery> tree;   // group 10: the nested boolean queries' descriptors (tq_tree.hip), one per query
  uint32_t total_tiles = 0, n_chunks = 0, max_k = 1;
  uint64_t list_entries = 0;  // term-major / doc-major groups: 8-byte entries of the group's result lists
  int kpl = 1;
  // offsets inside the staging blob
  size_t o_queries = 0, o_tiles = 0, o_outidx = 0, o_chunks = 0, o_perm = 0, o_sinks = 0;
  size_t o_leads = 0, o_tasks = 0, o_lists = 0;  // shared-union group
  void reset() {  // keeps the vectors' capacity
    queries.clear();
    out_index.clear();
    tile_starts.clear();
    chunk_recs.clear();
    tile_cost.clear();
    tree.clear();
    total_tiles = 0;
    n_chunks = 0;
    list_entries = 0;
    max_k = 1;
    kpl = 1;
    o_queries = o_tiles = o_outidx = o_chunks = o_perm = o_sinks = 0;
    o_leads = o_tasks = o_lists = 0;
  }
};

}  // namespace tqi

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
// (the shared-intersection planner orders the leads of one (leader, cache) by this bin of their mask: leads with the
// same mask are neighbours, two masks in one bin — 2 048 bins — may interleave; tools/planbench/plan_check.cpp)
constexpr uint32_t kALeadMaskBins = 2048;
inline uint32_t alead_mask_bin(uint64_t m) { return (uint32_t)((m * 0x9E3779B97F4A7C15ull) >> 53); }
struct ShareKey {  // one (query, list) pair of the shared-union group
  uint64_t key;    // list position i << 56 | blocks of the term (rare terms first) << 32 | cache
  uint32_t term, q;
};
// launch groups of a batch: 0 AND over bitmap lists, 1 unions, 2 phrases, 3 AND over any lists, 4 boolean
// queries, 5 shared unions, 6 phrase sweep, 7 doc-major unions, 8 shared intersections, 9 boolean queries
// through the shared-intersection launch
constexpr int kNGroups = 11;  // (10: nested boolean queries over bitmaps, tq_tree.hip)
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
  // shared-intersection launches (tq_ashare.hip): [0] the AND group (one lead per query), [1] the boolean
  // group (one lead per (query, list of its lead set)); leads sorted by (leader, cache, mask, query hash)
  struct ASharePlan {
    std::vector<TqdALead> aleads, aleads_unsorted;
    std::vector<ALeadKey> alead_keys, alead_keys2;
    std::vector<uint32_t> alead_first, alead_bucket, alead_bucket_at, alead_bucket_starts;
    std::vector<uint8_t> alead_same;
    std::vector<uint64_t> aq_sig;   // per query of the group: hash of the whole query (lists, weights, k, roles)
    std::vector<uint64_t> aq_slot;  // open-addressing table over aq_sig: the first query with that content
    struct QKey {
      uint32_t t0, t1, w0, w1, k, shape;
    };
    std::vector<QKey> aq_key;
    std::vector<uint32_t> aowner;  // per query of the group: the query whose result list it reads (itself, or the identical query before it)
    std::vector<uint4> atasks;
    std::vector<uint2> alists;  // boolean group: [query][list] = {bitmap, tf bytes} as offsets from the table base
    std::vector<uint32_t> apairs, atask_hist, atask_slab_run;
    struct ARun {  // the leads of one (leader, cache)
      uint32_t r0, r1, term, cache, n_blocks, n_groups, per_group, bpt, nb_warm, n_runs;
      size_t task0;
    };
    std::vector<ARun> aruns;
    uint32_t a_warm_tasks = 0;  // tasks [0, a_warm_tasks) are the warm-up launch
    bool over_budget = false;   // the last plan failed because its result lists exceed TQ_AS_LIST_MB at the longest tasks
    bool any_rdir = false;      // (boolean leads) some list of alists is probed through its range directory
  };
  ASharePlan ap[2];
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

namespace tqi {

// Planner threads (TQ_PLAN_THREADS, default 4, 1 = off): the chunk tables of a large batch are
// built in slabs of queries / slices / records.  The helpers are a process-wide pool of detached
// threads that sleep on a condition variable between jobs (created on first use, never torn
// down: a batch plans in four parallel steps, and spawning threads for each of them cost more
// than the steps themselves — 2.1 ms of host time per 10 000-query AND batch against 1.3 ms for
// the same tables built by one thread).  One job at a time: a caller that finds the pool busy
// (another segment planning on another thread) runs its slabs itself.
uint32_t plan_threads();  // TQ_PLAN_THREADS (default 1: the calling thread alone)
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
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
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
void parallel_slabs(uint32_t n_slabs, F &&fn) {  // fn(slab) for slab in [0, n_slabs)
  if (n_slabs <= 1) {
    if (n_slabs) fn(0u);
    return;
  }
  PlanPool::get().run(
      n_slabs, [](void *c, uint32_t i) { (*static_cast<typename std::remove_reference<F>::type *>(c))(i); }, (void *)&fn);
}

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

inline int kpl_for(uint32_t k) { return k <= 64 ? 1 : (k <= 128 ? 2 : (k <= 256 ? 4 : 16)); }

inline uint64_t xrow_key(uint32_t term, float w) {
  uint32_t wb;
  memcpy(&wb, &w, sizeof wb);
  return ((uint64_t)term << 32) | wb;
}
// Per-call execution options (tq_search_opts resolved against the segment's defaults): nothing
// a call needs is read from mutable segment state after this point.
struct CallOpts {
  bool exhaustive;
  float bound_slack;
  bool no_ashare = false, no_bshare = false;  // (internal) this call keeps intersections / boolean queries off the shared launch
};
// ---- tq_terms.cpp
void dense_arena_free(tq_segment *s);  // (tq_terms.cpp: the arena of the dense lists' side tables — a hipMalloc or a mapped address range)
int sync_terms(tq_segment *s, hipStream_t st);
void mark_term_dirty(tq_segment *s, uint32_t handle);
int build_flat(tq_segment *s, uint32_t handle, hipStream_t st, bool *ok);
int order_after_last_batch(tq_segment *s, hipStream_t st);
int wait_segment_idle(tq_segment *s);
// must: the caller cannot run without the tables (nested boolean queries): any segment size, the least recently used
// slot if none is free; else: only a free slot or one idle for a while
int build_probe_tables(tq_segment *s, uint32_t handle, bool *ok, bool must = false);
int build_probe_posdir(tq_segment *s, uint32_t handle, bool *ok);
void prep_apply_lmax(tq_segment *s, bool wait);    // rmax_list of lists prepared by finished tq_term_prepare_batch calls
void probe_begin_batch(tq_segment *s);             // a new batch is being planned (the pool's clock)
void probe_touch(tq_segment *s, uint32_t handle);  // the batch being planned uses the list's probe tables
int count_batch(tq_segment *s, const tq_query *queries, uint32_t n_queries, uint32_t *out_counts);
// a query as a bitwise expression over bitmap words (tq_count.cpp; checked on the CPU by tools/planbench/plan_check.cpp)
bool count_expression(tq_segment *s, const tq_query &q, TqkCountQuery &cq, bool &known, uint64_t &driver_postings,
                      std::unordered_map<uint32_t, uint32_t> &temp_slot, uint32_t max_temp);
// ---- the planners
int build_group_chunks(Group &g, bool or_windows, PlanScratch &ps, bool boolean_group = false);
int build_share_plan(tq_segment *s, Group &g, PlanScratch &ps);
int build_ashare_plan(tq_segment *s, Group &g, PlanScratch &ps, bool boolean = false);
int build_dense_plan(tq_segment *s, Group &g, PlanScratch &ps, uint32_t cus);
int plan_bool_query(tq_segment *s, const tq_query &q, uint32_t qi, TqdQuery &dq, uint64_t &qbytes,
                    uint32_t &n_tiles, uint32_t &tile_cost, uint32_t &n_thr_rows, bool exhaustive);
bool bool_query_is_tree(const tq_query &q);
int plan_tree_query(tq_segment *s, const tq_query &q, uint32_t qi, TqdTreeQuery &tq, uint64_t &qbytes, uint64_t table_base);
// ---- tq_search.cpp
int resolve_opts(const tq_segment *s, const tq_search_opts *o, CallOpts &co);
int search_batch_impl(tq_segment *s, const tq_query *queries, uint32_t n_queries, uint32_t out_stride,
                      float *d_out_scores, uint32_t *d_out_docs, uint32_t *d_out_counts, void *hip_stream,
                      const CallOpts &co);
int search_batch_host(tq_segment *s, const tq_query *queries, uint32_t n_queries, uint32_t out_stride,
                      float *out_scores, uint32_t *out_docs, uint32_t *out_counts, const CallOpts &co);
// The same in two halves (the submit queue: the next coalesced batch is planned and enqueued while this one runs):
// _begin plans and enqueues the batch on the segment's stream (caller holds the segment lock) — the merge kernels write
// scores | docs | counts into the slot's pinned host buffer — and records the slot's event; _end waits for that event
// (no lock needed) and leaves the rows readable at out.p, out.p + o_docs, out.p + o_counts.
struct HostBatchSlot {
  PinnedBuf out;
  hipEvent_t done = nullptr, done_blocking = nullptr;  // the batch's last kernel: waited for spinning / asleep
  bool blocking = false;                               // ... which of the two this batch recorded
  size_t o_docs = 0, o_counts = 0;
  uint32_t n = 0, stride = 0;
};
int search_batch_host_begin(tq_segment *s, const tq_query *queries, uint32_t n_queries, uint32_t out_stride,
                            const CallOpts &co, HostBatchSlot &slot);
int search_batch_host_end(tq_segment *s, HostBatchSlot &slot);

}  // namespace tqi
