// searcher.hpp — C++ host side above the C ABI, mirroring the slice of tantivy's public API that
// sits on the hot path: Searcher / Query / Weight / TopDocs (src/core/searcher.rs,
// src/query/{query,weight}.rs, src/query/boolean_query/boolean_query.rs,
// src/query/term_query/term_query.rs, src/query/phrase_query/phrase_query.rs,
// src/collector/top_score_collector.rs).  Same names, argument meaning and error behaviour;
// execution is delegated to the device through include/tantivy_amd.h.
//
// Not mirrored (out of scope): the term dictionary (FST), Directory/CompositeFile,
// schema, tokenizers.  A SegmentReader is therefore built from the field's raw sub-file bytes
// plus a term -> TermInfo table that the reference's TermDictionary would supply.
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/tantivy_amd.h"
#include "bm25.hpp"

namespace tantivy_amd {

// src/error.rs: the variants the hot path can produce
struct TantivyError : std::runtime_error {
  enum Kind { InvalidArgument, SchemaError, SystemError, DataCorruption, Unsupported } kind;
  TantivyError(Kind k, const std::string &msg) : std::runtime_error(msg), kind(k) {}
};

// src/postings/term_info.rs:10-17
struct TermInfo {
  uint32_t doc_freq = 0;
  uint64_t postings_start = 0, postings_end = 0;
  uint64_t positions_start = 0, positions_end = 0;
};

// src/lib.rs:337-344 — ordered by (segment_ord, doc_id)
struct DocAddress {
  uint32_t segment_ord = 0;
  DocId doc_id = 0;
  bool operator==(const DocAddress &o) const {
    return segment_ord == o.segment_ord && doc_id == o.doc_id;
  }
};

// A device-resident single-field segment (SegmentReader + InvertedIndexReader + FieldNormReader
// of the reference, for one indexed text field).
class SegmentReader {
 public:
  SegmentReader(tq_ctx *ctx, int device, uint32_t segment_ord, uint32_t max_doc,
                uint8_t record_option, const uint8_t *idx, size_t idx_len, const uint8_t *pos,
                size_t pos_len, const uint8_t *fieldnorm, size_t fn_len);
  // sub-files already resident on `device` (e.g. written by the device codec writers): copied
  // device to device, no host copy kept; total_num_tokens = the 8-byte header of the .idx sub-file
  struct DeviceResident {};
  SegmentReader(DeviceResident, tq_ctx *ctx, int device, uint32_t segment_ord, uint32_t max_doc,
                uint8_t record_option, const uint8_t *d_idx, size_t idx_len, const uint8_t *d_pos,
                size_t pos_len, const uint8_t *d_fieldnorm, size_t fn_len, uint64_t total_num_tokens);
  ~SegmentReader();
  SegmentReader(const SegmentReader &) = delete;
  SegmentReader &operator=(const SegmentReader &) = delete;

  void add_term(uint32_t term_id, const TermInfo &info);  // TermDictionary substitute
  // the segment's TermInfoStore: term ids are term ordinals, TermInfos are decoded on first use
  // (TermDictionary::term_info_from_ord, termdict.rs:185-188)
  void set_term_info_store(std::shared_ptr<const class TermInfoStore> store);
  const TermInfo *get_term_info(uint32_t term_id) const;  // None => nullptr
  tq_term_handle term_handle(uint32_t term_id);            // prepares on first use
  // the terms of a batch that no query has named yet, prepared TOGETHER (tq_term_prepare_batch: one staged upload,
  // one signature launch) — term_handle then finds every one of them
  void prepare_terms(const uint32_t *term_ids, size_t n);
  uint32_t max_doc() const { return max_doc_; }
  uint32_t segment_ord() const { return segment_ord_; }
  uint64_t total_num_tokens() const { return total_num_tokens_; }  // inverted_index_reader.rs:72-73
  uint8_t record_option() const { return record_option_; }
  tq_segment *raw() const { return seg_; }

 private:
  tq_segment *seg_ = nullptr;
  uint32_t segment_ord_, max_doc_;
  uint8_t record_option_;
  uint64_t total_num_tokens_ = 0;
  mutable std::mutex m_;  // term tables: Searcher::search may be called from many threads
  mutable std::unordered_map<uint32_t, TermInfo> terms_;
  std::unordered_map<uint32_t, tq_term_handle> handles_;
  // handles of small term ids, read without the lock (0xFFFFFFFE = not prepared yet): a 10 000-query batch names
  // 20 000 terms per collect_segment — a mutex and a hash lookup each were a third of a millisecond per step
  static constexpr uint32_t kFastHandles = 1u << 20;
  static constexpr tq_term_handle kHandleUnknown = 0xFFFFFFFEu;
  std::atomic<std::atomic<tq_term_handle> *> fast_handles_{nullptr};  // (allocated with the first prepared term, 4 MB)
  std::shared_ptr<const class TermInfoStore> store_;
};

enum class Occur { Should, Must, MustNot };  // src/query/occur.rs

// The query shapes the device path takes.  Anything else is reported as
// TantivyError::Unsupported so that the caller keeps tantivy's own CPU scorer for it
// (the `SpecializedScorer::Other` branch, boolean_weight.rs:595-597).
struct Query {
  enum Kind { Term, Boolean, Phrase } kind;
  // Term
  uint32_t term = 0;
  // Boolean: clauses of (Occur, sub query)
  std::vector<std::pair<Occur, Query>> clauses;
  size_t minimum_number_should_match = 0;  // boolean_query.rs:146-150
  // BoostQuery::new(this, boost) (boost_query.rs:14-31): BoostWeight::scorer hands
  // `boost * self.boost` down the tree (:70-72) and the leaves apply Bm25Weight::boost_by
  Score boost = 1.0f;
  // Phrase: (offset, term), slop 0 (phrase_query.rs:28-63)
  std::vector<std::pair<uint32_t, uint32_t>> phrase_terms;

  static Query term_query(uint32_t term) {
    Query q;
    q.kind = Term;
    q.term = term;
    return q;
  }
  static Query boolean(std::vector<std::pair<Occur, Query>> clauses) {
    Query q;
    q.kind = Boolean;
    q.clauses = std::move(clauses);
    return q;
  }
  Query &boosted(Score b) {  // BoostQuery::new(self, b): nested wrappers multiply (:70-72)
    boost *= b;
    return *this;
  }
  // BooleanQuery::set_minimum_number_should_match
  Query &set_minimum_number_should_match(size_t n) {
    minimum_number_should_match = n;
    return *this;
  }
  // PhraseQuery::new(terms): offsets 0..n
  static Query phrase(const std::vector<uint32_t> &terms) {
    Query q;
    q.kind = Phrase;
    for (uint32_t i = 0; i < terms.size(); ++i) q.phrase_terms.emplace_back(i, terms[i]);
    return q;
  }
  static Query phrase_with_offsets(std::vector<std::pair<uint32_t, uint32_t>> terms) {
    Query q;
    q.kind = Phrase;
    q.phrase_terms = std::move(terms);
    return q;
  }
};

// src/collector/top_score_collector.rs:61-96
class TopDocs {
 public:
  static TopDocs with_limit(size_t limit) {
    if (limit == 0) throw TantivyError(TantivyError::InvalidArgument, "Limit must be strictly greater than 0.");
    TopDocs t;
    t.limit_ = limit;
    return t;
  }
  TopDocs and_offset(size_t offset) const {
    TopDocs t = *this;
    t.offset_ = offset;
    return t;
  }
  TopDocs order_by_score() const { return *this; }
  size_t limit() const { return limit_; }
  size_t offset() const { return offset_; }

 private:
  size_t limit_ = 10, offset_ = 0;
};

using Fruit = std::vector<std::pair<Score, DocAddress>>;

// Query -> per-index Weight: executor choice + global BM25 statistics (Query::weight,
// src/query/query.rs:128-160; BooleanQuery::weight boolean_query.rs:157-169).
// A vector that keeps up to N elements inside the object (a Weight's term ids and weights: two heap allocations per
// query were most of Query::weight for a batch of 10 000 two-term queries, and made a second preparing thread six
// times slower than one), heap storage beyond that.  Trivially copyable element types only.
template <typename T, size_t N>
class InlineVec {
 public:
  InlineVec() = default;
  InlineVec(const InlineVec &o) { assign(o.begin(), o.end()); }
  InlineVec(InlineVec &&o) noexcept { steal(o); }
  InlineVec &operator=(const InlineVec &o) {
    if (this != &o) assign(o.begin(), o.end());
    return *this;
  }
  InlineVec &operator=(InlineVec &&o) noexcept {
    if (this != &o) {
      release();
      steal(o);
    }
    return *this;
  }
  InlineVec &operator=(std::initializer_list<T> il) {
    assign(il.begin(), il.end());
    return *this;
  }
  ~InlineVec() { release(); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T *data() { return heap_ ? heap_ : inl_; }
  const T *data() const { return heap_ ? heap_ : inl_; }
  T *begin() { return data(); }
  T *end() { return data() + n_; }
  const T *begin() const { return data(); }
  const T *end() const { return data() + n_; }
  T &operator[](size_t i) { return data()[i]; }
  const T &operator[](size_t i) const { return data()[i]; }
  void clear() { n_ = 0; }
  void push_back(const T &v) {
    grow(n_ + 1);
    data()[n_++] = v;
  }
  void resize(size_t n, const T &fill = T()) {
    grow(n);
    for (size_t i = n_; i < n; ++i) data()[i] = fill;
    n_ = n;
  }
  template <typename It>
  void assign(It first, It last) {
    const size_t n = (size_t)(last - first);
    n_ = 0;
    grow(n);
    T *d = data();
    for (size_t i = 0; i < n; ++i) d[i] = first[i];
    n_ = n;
  }

 private:
  void grow(size_t want) {
    if (want <= cap_) return;
    size_t cap = cap_ * 2;
    if (cap < want) cap = want;
    T *h = new T[cap];
    const T *d = data();
    for (size_t i = 0; i < n_; ++i) h[i] = d[i];
    delete[] heap_;
    heap_ = h;
    cap_ = cap;
  }
  void release() {
    delete[] heap_;
    heap_ = nullptr;
    cap_ = N;
    n_ = 0;
  }
  void steal(InlineVec &o) {
    n_ = o.n_;
    cap_ = o.cap_;
    heap_ = o.heap_;
    if (!heap_)
      for (size_t i = 0; i < n_; ++i) inl_[i] = o.inl_[i];
    o.heap_ = nullptr;
    o.cap_ = N;
    o.n_ = 0;
  }
  T inl_[N];
  T *heap_ = nullptr;
  size_t n_ = 0, cap_ = N;
};

struct Weight {
  uint8_t mode = TQ_MODE_AND;
  InlineVec<uint32_t, 4> terms;          // term ids in query order
  InlineVec<Score, 4> weights;           // per term (AND/OR) or one (phrase)
  std::vector<uint32_t> phrase_offsets;  // phrase
  std::vector<uint8_t> occurs;           // TQ_MODE_BOOL: enum tq_occur per term
  std::vector<uint8_t> clause_of;        // TQ_MODE_BOOL: clause index per term
  std::vector<uint8_t> nested_occurs;    // TQ_MODE_BOOL: the term's occur inside its clause (a nested BooleanQuery of
                                         // terms); empty = every clause is a term or a union of terms
  std::vector<uint8_t> atom_of;          // TQ_MODE_BOOL: member index of the term inside its clause (terms sharing one are
                                         // a nested intersection of terms); empty = every term its own member
  std::vector<uint8_t> clause_min_should;  // TQ_MODE_BOOL: the nested queries' minimum_number_should_match by clause
                                           // index (TQ_MAX_TERMS entries), or empty
  uint32_t min_should_match = 0;         // TQ_MODE_BOOL
  std::shared_ptr<Bm25Weight> bm25;      // holds the shared tf cache
};

class Searcher {
 public:
  explicit Searcher(std::vector<std::shared_ptr<SegmentReader>> segments);
  ~Searcher();
  Searcher(const Searcher &) = delete;
  Searcher &operator=(const Searcher &) = delete;
  // Bm25StatisticsProvider (bm25.rs:27-50)
  uint64_t total_num_docs() const;
  uint64_t total_num_tokens() const;
  uint64_t doc_freq(uint32_t term) const;
  size_t num_segments() const { return segments_.size(); }
  SegmentReader &segment_reader(size_t ord) { return *segments_[ord]; }

  // Statistics of segments held by OTHER ranks (one segment per GPU): the reference lets the
  // caller supply its own Bm25StatisticsProvider (bm25.rs:11-25) for exactly this.
  // NOT to be called while searches are in progress on this Searcher: the cached term weights are reset entry by
  // entry, and a concurrent weight() may store an idf computed from the statistics of before the call (tantivy
  // builds a new Searcher per reader reload; so does the host mirror's caller).
  void add_remote_statistics(uint64_t max_doc, uint64_t total_num_tokens,
                             const std::vector<std::pair<uint32_t, uint32_t>> &term_doc_freqs);

  Weight weight(const Query &query) const;
  // Query::weight of a BooleanQuery of unboosted term clauses that are all Must (TQ_MODE_AND) or all Should
  // (TQ_MODE_OR) — the same arithmetic as weight() without building the Query tree; term weights are cached
  // per Searcher (idf is a log per term per query otherwise)
  Weight weight_flat(uint8_t mode, const uint32_t *terms, uint32_t n_terms) const;
  // ... for a batch: the statistics and the field's tf cache are looked up once (FlatContext), not per query
  struct FlatContext {
    uint64_t nd = 0;
    std::shared_ptr<Bm25Weight> cache;
  };
  FlatContext flat_context() const;
  void weight_flat_into(const FlatContext &fc, uint8_t mode, const uint32_t *terms, uint32_t n_terms, Weight &w) const;
  // Searcher::search (searcher.rs:180-238) for one query / a batch of queries.  search() may be
  // called from any number of threads at once, like the reference's: the per-segment
  // collect_segment calls of concurrent searches are coalesced into batched launches (tq_search_one)
  Fruit search(const Query &query, const TopDocs &collector);
  std::vector<Fruit> search_batch(const std::vector<Weight> &weights, const TopDocs &collector);
  // Searcher::search(&query, &Count) for a batch (count_collector.rs:39-80: per-segment counts of
  // alive matching docs, summed by merge_fruits)
  std::vector<uint64_t> count_batch(const std::vector<Weight> &weights);
  // collect_segment for a batch on one segment: per-segment top-(offset+limit), sorted
  void collect_segment_batch(size_t segment_ord, const std::vector<Weight> &weights, uint32_t k,
                             std::vector<float> &scores, std::vector<uint32_t> &docs,
                             std::vector<uint32_t> &counts);
  // same, results left in device memory (enqueued on hip_stream, no host sync): feeds the
  // cross-rank all-gather of the one-segment-per-GPU deployment
  void collect_segment_batch_device(size_t segment_ord, const std::vector<Weight> &weights,
                                    uint32_t k, float *d_scores, uint32_t *d_docs,
                                    uint32_t *d_counts, void *hip_stream);

 private:
  // block-max metadata was selected under the segment's own average fieldnorm; with global
  // statistics the device widens those bounds by (1 + d)^2 (tq_common.hpp: block_max_score);
  // passed with every call (tq_search_opts)
  uint32_t bound_slack_ppm(const SegmentReader &seg) const;
  std::vector<std::shared_ptr<SegmentReader>> segments_;
  mutable std::mutex cache_m_;
  mutable std::shared_ptr<Bm25Weight> shared_cache_;  // one tf cache per field (avg fieldnorm)
  // idf * (1 + K1) of small term ids as float bits (0xFFFFFFFF = not computed yet), read without a lock; dropped
  // when the statistics change (add_remote_statistics)
  static constexpr uint32_t kFastWeights = 1u << 20;
  mutable std::atomic<std::atomic<uint32_t> *> fast_weights_{nullptr};
  Score term_weight_cached(uint32_t term, uint64_t nd) const;
  uint64_t remote_docs_ = 0, remote_tokens_ = 0;
  std::unordered_map<uint32_t, uint64_t> remote_doc_freq_;
};

}  // namespace tantivy_amd
