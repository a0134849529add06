// searcher.cpp — see searcher.hpp.
#include "searcher.hpp"

#include "term_info_store.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace tantivy_amd {

namespace {
[[noreturn]] void throw_tq(int code) {
  const std::string msg = tq_last_error();
  switch (code) {
    case TQ_ERR_INVALID: throw TantivyError(TantivyError::InvalidArgument, msg);
    case TQ_ERR_FORMAT: throw TantivyError(TantivyError::DataCorruption, msg);
    case TQ_ERR_UNSUPPORTED: throw TantivyError(TantivyError::Unsupported, msg);
    default: throw TantivyError(TantivyError::SystemError, msg);
  }
}
}  // namespace

SegmentReader::SegmentReader(tq_ctx *ctx, int device, uint32_t segment_ord, uint32_t max_doc,
                             uint8_t record_option, const uint8_t *idx, size_t idx_len,
                             const uint8_t *pos, size_t pos_len, const uint8_t *fieldnorm,
                             size_t fn_len)
    : segment_ord_(segment_ord), max_doc_(max_doc), record_option_(record_option) {
  const int rc = tq_segment_upload(ctx, device, max_doc, idx, idx_len, pos, pos_len, fieldnorm,
                                   fn_len, record_option, &seg_);
  if (rc != TQ_OK) throw_tq(rc);
  uint64_t t = 0;
  for (int i = 0; i < 8; ++i) t |= (uint64_t)idx[i] << (8 * i);
  total_num_tokens_ = t;
}
SegmentReader::SegmentReader(DeviceResident, tq_ctx *ctx, int device, uint32_t segment_ord,
                             uint32_t max_doc, uint8_t record_option, const uint8_t *d_idx,
                             size_t idx_len, const uint8_t *d_pos, size_t pos_len,
                             const uint8_t *d_fieldnorm, size_t fn_len, uint64_t total_num_tokens)
    : segment_ord_(segment_ord), max_doc_(max_doc), record_option_(record_option),
      total_num_tokens_(total_num_tokens) {
  const int rc = tq_segment_upload_device(ctx, device, max_doc, d_idx, idx_len, d_pos, pos_len,
                                          d_fieldnorm, fn_len, record_option, &seg_);
  if (rc != TQ_OK) throw_tq(rc);
}
SegmentReader::~SegmentReader() {
  tq_segment_free(seg_);
  delete[] fast_handles_.load(std::memory_order_relaxed);
}

void SegmentReader::add_term(uint32_t term_id, const TermInfo &info) { terms_[term_id] = info; }
void SegmentReader::set_term_info_store(std::shared_ptr<const TermInfoStore> store) {
  store_ = std::move(store);
}
const TermInfo *SegmentReader::get_term_info(uint32_t term_id) const {
  std::lock_guard<std::mutex> lk(m_);  // (entries are never erased: the pointer stays valid)
  auto it = terms_.find(term_id);
  if (it == terms_.end() && store_ && term_id < store_->num_terms())
    it = terms_.emplace(term_id, store_->get(term_id)).first;
  if (it == terms_.end() || it->second.doc_freq == 0) return nullptr;
  return &it->second;
}
tq_term_handle SegmentReader::term_handle(uint32_t term_id) {
  if (term_id < kFastHandles) {  // (the table exists from the first prepared term on)
    std::atomic<tq_term_handle> *fast = fast_handles_.load(std::memory_order_acquire);
    if (fast) {
      const tq_term_handle h = fast[term_id].load(std::memory_order_acquire);
      if (h != kHandleUnknown) return h;
    }
  }
  {
    std::lock_guard<std::mutex> lk(m_);
    auto h = handles_.find(term_id);
    if (h != handles_.end()) return h->second;
  }
  const TermInfo *ti = get_term_info(term_id);
  tq_term_handle handle = TQ_TERM_ABSENT;
  if (ti) {
    const int rc = tq_term_prepare(seg_, ti->postings_start,
                                   (uint32_t)(ti->postings_end - ti->postings_start),
                                   ti->positions_start,
                                   (uint32_t)(ti->positions_end - ti->positions_start),
                                   ti->doc_freq, &handle);
    if (rc != TQ_OK) throw_tq(rc);  // (tq_term_prepare is idempotent per postings offset: two threads
  }                                 // preparing the same term get the same handle)
  std::lock_guard<std::mutex> lk(m_);
  handles_[term_id] = handle;
  if (term_id < kFastHandles) {
    std::atomic<tq_term_handle> *fast = fast_handles_.load(std::memory_order_relaxed);
    if (!fast) {
      fast = new std::atomic<tq_term_handle>[kFastHandles];
      for (uint32_t i = 0; i < kFastHandles; ++i) fast[i].store(kHandleUnknown, std::memory_order_relaxed);
      fast_handles_.store(fast, std::memory_order_release);  // (readers that still see null take the locked path)
    }
    fast[term_id].store(handle, std::memory_order_release);
  }
  return handle;
}

void SegmentReader::prepare_terms(const uint32_t *term_ids, size_t n) {
  // the ones without a handle yet, each once
  std::vector<uint32_t> fresh;
  {
    std::atomic<tq_term_handle> *fast = fast_handles_.load(std::memory_order_acquire);
    std::unordered_map<uint32_t, char> seen;
    for (size_t i = 0; i < n; ++i) {
      const uint32_t t = term_ids[i];
      if (t < kFastHandles && fast && fast[t].load(std::memory_order_acquire) != kHandleUnknown) continue;
      if (!seen.emplace(t, 1).second) continue;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (handles_.count(t)) continue;
      }
      fresh.push_back(t);
    }
  }
  if (fresh.size() < 2) return;  // (one term: term_handle's own road)
  std::vector<tq_term_info> infos;
  std::vector<uint32_t> ids;
  for (uint32_t t : fresh) {
    const TermInfo *ti = get_term_info(t);
    if (!ti) continue;  // (absent: term_handle records TQ_TERM_ABSENT)
    tq_term_info x{};
    x.postings_off = ti->postings_start;
    x.postings_len = (uint32_t)(ti->postings_end - ti->postings_start);
    x.positions_off = ti->positions_start;
    x.positions_len = (uint32_t)(ti->positions_end - ti->positions_start);
    x.doc_freq = ti->doc_freq;
    infos.push_back(x);
    ids.push_back(t);
  }
  if (infos.empty()) return;
  std::vector<tq_term_handle> hs(infos.size(), TQ_TERM_ABSENT);
  const int rc = tq_term_prepare_batch(seg_, infos.data(), (uint32_t)infos.size(), hs.data());
  if (rc != TQ_OK) throw_tq(rc);
  std::lock_guard<std::mutex> lk(m_);
  std::atomic<tq_term_handle> *fast = fast_handles_.load(std::memory_order_relaxed);
  if (!fast) {
    fast = new std::atomic<tq_term_handle>[kFastHandles];
    for (uint32_t i = 0; i < kFastHandles; ++i) fast[i].store(kHandleUnknown, std::memory_order_relaxed);
    fast_handles_.store(fast, std::memory_order_release);
  }
  for (size_t i = 0; i < ids.size(); ++i) {
    handles_[ids[i]] = hs[i];
    if (ids[i] < kFastHandles) fast[ids[i]].store(hs[i], std::memory_order_release);
  }
}

Searcher::Searcher(std::vector<std::shared_ptr<SegmentReader>> segments)
    : segments_(std::move(segments)) {}
Searcher::~Searcher() { delete[] fast_weights_.load(std::memory_order_relaxed); }

void Searcher::add_remote_statistics(
    uint64_t max_doc, uint64_t total_num_tokens,
    const std::vector<std::pair<uint32_t, uint32_t>> &term_doc_freqs) {
  remote_docs_ += max_doc;
  remote_tokens_ += total_num_tokens;
  for (auto &tf : term_doc_freqs) remote_doc_freq_[tf.first] += tf.second;
  shared_cache_.reset();
  if (std::atomic<uint32_t> *fw = fast_weights_.load(std::memory_order_acquire))  // (the statistics changed)
    for (uint32_t i = 0; i < kFastWeights; ++i) fw[i].store(0xFFFFFFFFu, std::memory_order_relaxed);
}

Score Searcher::term_weight_cached(uint32_t term, uint64_t nd) const {
  std::atomic<uint32_t> *fw = term < kFastWeights ? fast_weights_.load(std::memory_order_acquire) : nullptr;
  if (fw) {
    const uint32_t bits = fw[term].load(std::memory_order_relaxed);
    if (bits != 0xFFFFFFFFu) {
      Score w;
      std::memcpy(&w, &bits, sizeof w);
      return w;
    }
  }
  const Score w = idf(doc_freq(term), nd) * (1.0f + K1);  // TermQuery::specialized_weight -> Bm25Weight::for_terms
  if (term < kFastWeights) {
    if (!fw) {
      std::lock_guard<std::mutex> lk(cache_m_);
      fw = fast_weights_.load(std::memory_order_relaxed);
      if (!fw) {
        fw = new std::atomic<uint32_t>[kFastWeights];
        for (uint32_t i = 0; i < kFastWeights; ++i) fw[i].store(0xFFFFFFFFu, std::memory_order_relaxed);
        fast_weights_.store(fw, std::memory_order_release);
      }
    }
    uint32_t bits;
    std::memcpy(&bits, &w, sizeof bits);
    fw[term].store(bits, std::memory_order_relaxed);
  }
  return w;
}

Searcher::FlatContext Searcher::flat_context() const {
  FlatContext fc;
  fc.nd = total_num_docs();
  const uint64_t nt = total_num_tokens();
  if (fc.nd == 0)
    throw TantivyError(TantivyError::InvalidArgument, "no documents: BM25 statistics undefined");
  std::lock_guard<std::mutex> lk(cache_m_);
  if (!shared_cache_) {
    const Score avg = (Score)nt / (Score)fc.nd;
    shared_cache_ = std::make_shared<Bm25Weight>(Bm25Weight::from_idf(0.0f, avg));
  }
  fc.cache = shared_cache_;
  return fc;
}
void Searcher::weight_flat_into(const FlatContext &fc, uint8_t mode, const uint32_t *terms, uint32_t n_terms, Weight &w) const {
  if (n_terms > TQ_MAX_TERMS)
    throw TantivyError(TantivyError::Unsupported, "more than 16 terms stay on the CPU");
  w.bm25 = fc.cache;
  w.mode = n_terms == 1 ? (uint8_t)TQ_MODE_OR : mode;  // (a one-term query: TermWeight::for_each_pruning)
  w.terms.assign(terms, terms + n_terms);
  w.weights.resize(n_terms);
  for (uint32_t i = 0; i < n_terms; ++i) w.weights[i] = term_weight_cached(terms[i], fc.nd);
}
Weight Searcher::weight_flat(uint8_t mode, const uint32_t *terms, uint32_t n_terms) const {
  Weight w;
  weight_flat_into(flat_context(), mode, terms, n_terms, w);
  return w;
}
uint64_t Searcher::total_num_docs() const {
  uint64_t n = remote_docs_;
  for (auto &s : segments_) n += s->max_doc();
  return n;
}
uint64_t Searcher::total_num_tokens() const {
  uint64_t n = remote_tokens_;
  for (auto &s : segments_) n += s->total_num_tokens();
  return n;
}
uint64_t Searcher::doc_freq(uint32_t term) const {
  uint64_t n = 0;
  auto it = remote_doc_freq_.find(term);
  if (it != remote_doc_freq_.end()) n = it->second;
  for (auto &s : segments_)
    if (const TermInfo *ti = s->get_term_info(term)) n += ti->doc_freq;
  return n;
}

Weight Searcher::weight(const Query &query) const {
  const uint64_t nd = total_num_docs(), nt = total_num_tokens();
  if (nd == 0)
    throw TantivyError(TantivyError::InvalidArgument, "no documents: BM25 statistics undefined");
  Weight w;
  {
    std::lock_guard<std::mutex> lk(cache_m_);
    if (!shared_cache_) {
      const Score avg = (Score)nt / (Score)nd;
      shared_cache_ = std::make_shared<Bm25Weight>(Bm25Weight::from_idf(0.0f, avg));
    }
    w.bm25 = shared_cache_;
  }
  // Weight::scorer(reader, boost) starts at 1.0; every BoostWeight multiplies its own factor in
  // (boost_query.rs:70-72) and the leaf applies Bm25Weight::boost_by(boost) (bm25.rs:82-92)
  auto boost_by = [](Score weight, Score boost) { return boost == 1.0f ? weight : weight * boost; };
  const Score b0 = 1.0f * query.boost;
  auto term_weight = [&](uint32_t term, Score boost) {
    // TermQuery::specialized_weight -> Bm25Weight::for_terms(statistics, [term])
    return boost_by(idf(doc_freq(term), nd) * (1.0f + K1), boost);
  };
  switch (query.kind) {
    case Query::Term:
      // TermWeight::for_each_pruning == one-scorer block-WAND == every doc of the list
      w.mode = TQ_MODE_OR;
      w.terms = {query.term};
      w.weights = {term_weight(query.term, b0)};
      return w;
    case Query::Phrase: {
      if (query.phrase_terms.size() < 2)
        throw TantivyError(TantivyError::InvalidArgument,
                           "A phrase query is required to have strictly more than one term.");
      w.mode = TQ_MODE_PHRASE;
      Score idf_sum = 0.0f;  // bm25.rs:121-128
      for (auto &ot : query.phrase_terms) {
        w.phrase_offsets.push_back(ot.first);
        w.terms.push_back(ot.second);
        idf_sum += idf(doc_freq(ot.second), nd);
      }
      w.weights = {boost_by(idf_sum * (1.0f + K1), b0)};
      return w;
    }
    case Query::Boolean: {
      // BooleanWeight::complex_scorer (boolean_weight.rs:236-431).  Clauses are terms or nested
      // BooleanQuerys of Should terms (`+a +(b OR c)`): all Must terms -> TermIntersection
      // (block_wand_intersection), all Should terms -> TermUnion (block_wand), anything else ->
      // Intersection / RequiredOptionalScorer / Exclude / Disjunction on the union kernel.
      if (query.clauses.empty())
        throw TantivyError(TantivyError::Unsupported, "empty boolean query");
      // A Must clause that is itself a BooleanQuery with at least one Must term (`+a +(+b +c)`,
      // `+a +(+b -c)`, `+a +(+b c)`) is an Intersection of the parent's required scorers with the
      // nested query's scorer (boolean_weight.rs:308-431): the same doc set and the same score
      // terms as the nested clauses hoisted into the parent — required stays required, excluded
      // stays excluded, and the nested optional clauses stay optional because the parent has a Must.
      // Only the association of the f32 sum differs (a + (b + c) against a + b + c): within the
      // 1e-5 the reference's own cursor-order sums already move by.  Hoisting needs
      // minimum_number_should_match == 0 on both levels (its count is per BooleanQuery).
      std::vector<std::pair<Occur, Query>> hoisted;
      const std::vector<std::pair<Occur, Query>> *clauses_p = &query.clauses;
      {
        auto hoistable = [](const std::pair<Occur, Query> &c) {
          if (c.first != Occur::Must || c.second.kind != Query::Boolean ||
              c.second.minimum_number_should_match != 0 || c.second.clauses.empty())
            return false;
          bool has_must = false;
          for (auto &sc : c.second.clauses) {
            if (sc.second.kind != Query::Term) return false;
            has_must |= sc.first == Occur::Must;
          }
          return has_must;
        };
        bool any = false;
        for (auto &c : query.clauses) any |= hoistable(c);
        if (any && query.minimum_number_should_match == 0) {
          for (auto &c : query.clauses) {
            if (!hoistable(c)) {
              hoisted.push_back(c);
              continue;
            }
            for (auto &sc : c.second.clauses) {
              Query t = sc.second;
              t.boost *= c.second.boost;  // BoostQuery around the nested query reaches its leaves
              hoisted.emplace_back(sc.first, std::move(t));
            }
          }
          clauses_p = &hoisted;
        }
      }
      const std::vector<std::pair<Occur, Query>> &clauses = *clauses_p;
      bool all_must = true, all_should = true, flat = true, tree = false;
      // a clause is a term, a union of terms, or (round 5) any BooleanQuery of TERMS — an intersection inside a
      // union or under MustNot, nested MustNot / optional terms, a nested minimum_number_should_match: depth 2,
      // evaluated over the lists' bitmaps (tq_tree.hip).  Deeper trees keep
      // tantivy's CPU scorer (SpecializedScorer::Other, boolean_weight.rs:595-597).
      // (a member of a nested query: a term, or an intersection of terms — `(+b +c) d`)
      auto is_conjunction = [](const Query &q) {
        if (q.kind != Query::Boolean || q.clauses.empty() || q.minimum_number_should_match != 0) return false;
        for (auto &c : q.clauses)
          if (c.first != Occur::Must || c.second.kind != Query::Term) return false;
        return true;
      };
      // (round 5, second half: a member — or a clause — can be a PhraseQuery of <= 4 terms: PhraseScorer under
      // Intersection / union / Exclude, the position check done per candidate doc on the device)
      bool any_phrase = false;
      // (round 6: ... or a UNION of terms one level down — `(+(b c) +d) e`: BufferedUnionScorer as a member)
      auto is_member_union = [](const Query &q) {
        if (q.kind != Query::Boolean || q.clauses.size() < 2 || q.minimum_number_should_match > 1) return false;
        for (auto &c : q.clauses)
          if (c.first != Occur::Should || c.second.kind != Query::Term) return false;
        return true;
      };
      auto is_query_of_terms = [&](const Query &q) {
        if (q.kind != Query::Boolean || q.clauses.empty()) return false;
        for (auto &c : q.clauses)
          if (c.second.kind != Query::Term && c.second.kind != Query::Phrase && !is_conjunction(c.second) &&
              !is_member_union(c.second))
            return false;
        return true;
      };
      auto is_term_union = [](const Query &q) {
        if (q.minimum_number_should_match > 1) return false;
        for (auto &c : q.clauses)
          if (c.first != Occur::Should || c.second.kind != Query::Term) return false;
        return true;
      };
      for (auto &c : clauses) {
        if (c.second.kind == Query::Phrase) {
          any_phrase = tree = true;
          flat = false;
        } else if (c.second.kind != Query::Term) {
          if (!is_query_of_terms(c.second))
            throw TantivyError(TantivyError::Unsupported,
                               "boolean trees deeper than three levels (or with a nested query below the second) stay on the CPU scorer path");
          for (auto &sub : c.second.clauses) any_phrase = any_phrase || sub.second.kind == Query::Phrase;
          tree = tree || any_phrase || !is_term_union(c.second);
          flat = false;
        }
        all_must &= c.first == Occur::Must;
        all_should &= c.first == Occur::Should;
      }
      const size_t msm = query.minimum_number_should_match;
      if (tree)
        w.mode = TQ_MODE_BOOL;
      else if (flat && all_must && msm == 0)
        w.mode = TQ_MODE_AND;
      else if (flat && all_should && msm <= 1)
        w.mode = TQ_MODE_OR;
      else
        w.mode = TQ_MODE_BOOL;
      uint8_t clause = 0;
      for (auto &c : clauses) {
        const uint8_t oc = c.first == Occur::Must
                               ? (uint8_t)TQ_MUST
                               : (c.first == Occur::MustNot ? (uint8_t)TQ_MUST_NOT
                                                            : (uint8_t)TQ_SHOULD);
        const Score bc = b0 * c.second.boost;  // the clause's own BoostQuery wrapper, if any
        uint8_t member = 0;
        auto add = [&](uint32_t term, Score boost, Occur inner, uint8_t flag = 0) {
          w.terms.push_back(term);
          w.weights.push_back(term_weight(term, boost));
          if (w.mode == TQ_MODE_BOOL) {
            w.occurs.push_back(oc);
            w.clause_of.push_back(clause);
            if (tree) {
              w.nested_occurs.push_back((uint8_t)((inner == Occur::Must ? (uint8_t)TQ_MUST
                                                                         : (inner == Occur::MustNot ? (uint8_t)TQ_MUST_NOT : (uint8_t)TQ_SHOULD)) | flag));
              w.atom_of.push_back(member);
            }
          }
        };
        // a PhraseQuery as one member: its terms share the member index, carry TQ_NESTED_PHRASE, their offsets
        // and — each — the phrase's weight (Bm25Weight::for_terms over the phrase's terms, bm25.rs:121-128)
        auto add_phrase = [&](const Query &ph, Score boost, Occur inner) {
          if (ph.phrase_terms.size() < 2)
            throw TantivyError(TantivyError::InvalidArgument,
                               "A phrase query is required to have strictly more than one term.");
          if (ph.phrase_terms.size() > 8)
            throw TantivyError(TantivyError::Unsupported, "a phrase inside a boolean query takes at most 8 terms on the device");
          Score idf_sum = 0.0f;
          for (auto &ot : ph.phrase_terms) idf_sum += idf(doc_freq(ot.second), nd);
          const Score pw = boost_by(idf_sum * (1.0f + K1), boost);
          for (auto &ot : ph.phrase_terms) {
            w.phrase_offsets.resize(w.terms.size(), 0u);
            w.phrase_offsets.push_back(ot.first);
            w.terms.push_back(ot.second);
            w.weights.push_back(pw);
            w.occurs.push_back(oc);
            w.clause_of.push_back(clause);
            w.nested_occurs.push_back((uint8_t)((inner == Occur::Must ? TQ_MUST : (inner == Occur::MustNot ? TQ_MUST_NOT : TQ_SHOULD)) |
                                                TQ_NESTED_PHRASE));
            w.atom_of.push_back(member);
          }
        };
        if (c.second.kind == Query::Term) {
          add(c.second.term, bc, Occur::Must);  // (a one-term clause: the term itself)
        } else if (c.second.kind == Query::Phrase) {
          add_phrase(c.second, bc, Occur::Must);  // (a clause of its own: a one-member nested query)
        } else {
          for (auto &sub : c.second.clauses) {
            if (sub.second.kind == Query::Term)
              add(sub.second.term, bc * sub.second.boost, sub.first);
            else if (sub.second.kind == Query::Phrase)
              add_phrase(sub.second, bc * sub.second.boost, sub.first);
            else if (is_member_union(sub.second))  // a union of terms one level down: one member (TQ_NESTED_ANY)
              for (auto &leaf : sub.second.clauses) add(leaf.second.term, bc * sub.second.boost * leaf.second.boost, sub.first, TQ_NESTED_ANY);
            else  // an intersection of terms one level down: one member of the nested query
              for (auto &leaf : sub.second.clauses) add(leaf.second.term, bc * sub.second.boost * leaf.second.boost, sub.first);
            ++member;
          }
          if (tree && c.second.minimum_number_should_match) {
            if (clause >= TQ_MAX_TERMS || c.second.minimum_number_should_match > 255)
              throw TantivyError(TantivyError::Unsupported, "nested minimum_number_should_match out of range");
            if (w.clause_min_should.empty()) w.clause_min_should.assign(TQ_MAX_TERMS, 0);
            w.clause_min_should[clause] = (uint8_t)c.second.minimum_number_should_match;
          }
        }
        ++clause;
      }
      if (w.terms.size() > TQ_MAX_TERMS)
        throw TantivyError(TantivyError::Unsupported, "more than 16 terms stay on the CPU");
      if (any_phrase) w.phrase_offsets.resize(w.terms.size(), 0u);
      w.min_should_match = (uint32_t)std::min<size_t>(msm, 0xFFFFu);
      return w;
    }
  }
  throw TantivyError(TantivyError::InvalidArgument, "unknown query kind");
}

namespace {
// builds the tq_query array of a batch for one segment (scorer construction of
// Weight::scorer per segment: term lookups only, the device owns the rest)
struct SegmentBatch {
  std::vector<tq_query> qs;
  std::vector<tq_term_handle> handles;
  SegmentBatch(SegmentReader &seg, const std::vector<Weight> &weights, uint32_t k) {
    const size_t n = weights.size();
    qs.resize(n);
    size_t total_terms = 0;
    for (auto &w : weights) total_terms += w.terms.size();
    handles.reserve(total_terms);
    if (n > 1) {  // the batch's new terms are prepared together before any handle is asked for
      std::vector<uint32_t> all;
      all.reserve(total_terms);
      for (auto &w : weights) all.insert(all.end(), w.terms.begin(), w.terms.end());
      seg.prepare_terms(all.data(), all.size());
    }
    for (size_t i = 0; i < n; ++i) {
      const Weight &w = weights[i];
      const size_t at = handles.size();
      for (uint32_t t : w.terms) handles.push_back(seg.term_handle(t));
      tq_query &q = qs[i];
      q.n_terms = (uint32_t)w.terms.size();
      q.terms = handles.data() + at;  // stable: reserved up front
      q.weights = w.weights.data();
      q.tf_cache = w.bm25->cache;
      q.mode = w.mode;
      q.phrase_offsets = w.phrase_offsets.empty() ? nullptr : w.phrase_offsets.data();
      q.k = k;
      q.occurs = w.occurs.empty() ? nullptr : w.occurs.data();
      q.clause_of = w.clause_of.empty() ? nullptr : w.clause_of.data();
      q.min_should_match = w.min_should_match;
      q.nested_occurs = w.nested_occurs.empty() ? nullptr : w.nested_occurs.data();
      q.clause_min_should = w.clause_min_should.empty() ? nullptr : w.clause_min_should.data();
      q.atom_of = w.atom_of.empty() ? nullptr : w.atom_of.data();
    }
  }
};
}  // namespace

uint32_t Searcher::bound_slack_ppm(const SegmentReader &seg) const {
  const double global = (double)total_num_tokens() / (double)total_num_docs();
  const double local = seg.max_doc() ? (double)seg.total_num_tokens() / (double)seg.max_doc() : global;
  double ppm = 0.0;
  if (global > 0.0 && local > 0.0 && global != local) {
    const double d = std::abs(global - local) / std::min(global, local);
    ppm = std::ceil(((1.0 + d) * (1.0 + d) - 1.0) * 1e6) + 2.0;  // + f32 rounding of the averages
  }
  return (uint32_t)std::min(ppm, 1e9);
}

void Searcher::collect_segment_batch(size_t segment_ord, const std::vector<Weight> &weights,
                                     uint32_t k, std::vector<float> &scores,
                                     std::vector<uint32_t> &docs, std::vector<uint32_t> &counts) {
  SegmentReader &seg = *segments_[segment_ord];
  const size_t n = weights.size();
  SegmentBatch b(seg, weights, k);
  scores.assign(n * k, 0.0f);
  docs.assign(n * k, TERMINATED);
  counts.assign(n, 0);
  // the slack travels with the call: segments shared by several Searchers never race on it
  const tq_search_opts opts{-1, bound_slack_ppm(seg)};
  const int rc = tq_search_batch_opts(seg.raw(), b.qs.data(), (uint32_t)n, k, scores.data(),
                                      docs.data(), counts.data(), &opts);
  if (rc != TQ_OK) throw_tq(rc);
}

std::vector<uint64_t> Searcher::count_batch(const std::vector<Weight> &weights) {
  std::vector<uint64_t> total(weights.size(), 0);
  std::vector<uint32_t> per(weights.size());
  for (auto &seg : segments_) {
    SegmentBatch b(*seg, weights, 1);
    const int rc = tq_count_batch(seg->raw(), b.qs.data(), (uint32_t)weights.size(), per.data());
    if (rc != TQ_OK) throw_tq(rc);
    for (size_t i = 0; i < per.size(); ++i) total[i] += per[i];
  }
  return total;
}

void Searcher::collect_segment_batch_device(size_t segment_ord, const std::vector<Weight> &weights,
                                            uint32_t k, float *d_scores, uint32_t *d_docs,
                                            uint32_t *d_counts, void *hip_stream) {
  SegmentReader &seg = *segments_[segment_ord];
  SegmentBatch b(seg, weights, k);
  const tq_search_opts opts{-1, bound_slack_ppm(seg)};
  const int rc = tq_search_batch_device_opts(seg.raw(), b.qs.data(), (uint32_t)weights.size(), k,
                                             d_scores, d_docs, d_counts, &opts, hip_stream);
  if (rc != TQ_OK) throw_tq(rc);
}

std::vector<Fruit> Searcher::search_batch(const std::vector<Weight> &weights,
                                          const TopDocs &collector) {
  const size_t n = weights.size(), S = segments_.size();
  const uint32_t k = (uint32_t)(collector.offset() + collector.limit());  // collect_segment: k = doc_range.end
  if (k > TQ_MAX_K)
    throw TantivyError(TantivyError::Unsupported, "offset+limit above the device heap size");
  std::vector<float> all_scores(S * n * k);
  std::vector<uint32_t> all_docs(S * n * k), all_counts(S * n);
  std::vector<float> sc;
  std::vector<uint32_t> dc, ct;
  for (size_t s = 0; s < S; ++s) {  // Executor::SingleThread order (executor.rs:52-59)
    collect_segment_batch(s, weights, k, sc, dc, ct);
    std::memcpy(all_scores.data() + s * n * k, sc.data(), n * k * sizeof(float));
    std::memcpy(all_docs.data() + s * n * k, dc.data(), n * k * sizeof(uint32_t));
    std::memcpy(all_counts.data() + s * n, ct.data(), n * sizeof(uint32_t));
  }
  const uint32_t limit = (uint32_t)collector.limit(), offset = (uint32_t)collector.offset();
  std::vector<float> out_s(n * limit);
  std::vector<uint32_t> out_o(n * limit), out_d(n * limit), out_c(n);
  const int rc = tq_merge_topk(all_scores.data(), all_docs.data(), all_counts.data(), (uint32_t)S,
                               (uint32_t)n, k, offset, limit, out_s.data(), out_o.data(),
                               out_d.data(), out_c.data());
  if (rc != TQ_OK) throw_tq(rc);
  std::vector<Fruit> res(n);
  for (size_t q = 0; q < n; ++q) {
    res[q].reserve(out_c[q]);
    for (uint32_t i = 0; i < out_c[q]; ++i)
      res[q].push_back({out_s[q * limit + i],
                        DocAddress{segments_[out_o[q * limit + i]]->segment_ord(),
                                   out_d[q * limit + i]}});
  }
  return res;
}

Fruit Searcher::search(const Query &query, const TopDocs &collector) {
  // Searcher::search_with_executor (searcher.rs:215-238): the weight once, then collect_segment per
  // segment, then merge_fruits.  Callable from many threads: each collect_segment is a tq_search_one,
  // which rides in whatever batch the segment launches next.
  const Weight w = weight(query);
  const size_t S = segments_.size();
  const uint32_t k = (uint32_t)(collector.offset() + collector.limit());
  if (k > TQ_MAX_K)
    throw TantivyError(TantivyError::Unsupported, "offset+limit above the device heap size");
  std::vector<float> all_scores(S * k);
  std::vector<uint32_t> all_docs(S * k), all_counts(S);
  std::vector<tq_term_handle> handles(w.terms.size());
  for (size_t s = 0; s < S; ++s) {
    SegmentReader &seg = *segments_[s];
    for (size_t i = 0; i < w.terms.size(); ++i) handles[i] = seg.term_handle(w.terms[i]);
    tq_query q{};
    q.n_terms = (uint32_t)w.terms.size();
    q.terms = handles.data();
    q.weights = w.weights.data();
    q.tf_cache = w.bm25->cache;
    q.mode = w.mode;
    q.phrase_offsets = w.phrase_offsets.empty() ? nullptr : w.phrase_offsets.data();
    q.k = k;
    q.occurs = w.occurs.empty() ? nullptr : w.occurs.data();
    q.clause_of = w.clause_of.empty() ? nullptr : w.clause_of.data();
    q.min_should_match = w.min_should_match;
    q.nested_occurs = w.nested_occurs.empty() ? nullptr : w.nested_occurs.data();
    q.clause_min_should = w.clause_min_should.empty() ? nullptr : w.clause_min_should.data();
    q.atom_of = w.atom_of.empty() ? nullptr : w.atom_of.data();
    const tq_search_opts opts{-1, bound_slack_ppm(seg)};
    const int rc = tq_search_one(seg.raw(), &q, &opts, all_scores.data() + s * k, all_docs.data() + s * k,
                                 all_counts.data() + s);
    if (rc != TQ_OK) throw_tq(rc);
  }
  const uint32_t limit = (uint32_t)collector.limit(), offset = (uint32_t)collector.offset();
  std::vector<float> out_s(limit);
  std::vector<uint32_t> out_o(limit), out_d(limit);
  uint32_t out_c = 0;
  const int rc = tq_merge_topk(all_scores.data(), all_docs.data(), all_counts.data(), (uint32_t)S, 1u, k, offset,
                               limit, out_s.data(), out_o.data(), out_d.data(), &out_c);
  if (rc != TQ_OK) throw_tq(rc);
  Fruit res;
  res.reserve(out_c);
  for (uint32_t i = 0; i < out_c; ++i)
    res.push_back({out_s[i], DocAddress{segments_[out_o[i]]->segment_ord(), out_d[i]}});
  return res;
}

}  // namespace tantivy_amd
