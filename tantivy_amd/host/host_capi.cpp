// host_capi.cpp — flat C entry points over the C++ host mirror (searcher.hpp), so that the
// Python tests/bench drive exactly the host code a C++ application would.  Exceptions never
// cross this boundary: they become status codes + tqh_last_error().
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstring>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "searcher.hpp"
#include "term_info_store.hpp"

using namespace tantivy_amd;

namespace {
thread_local std::string g_err;
template <typename F>
int guard(F &&f) {
  try {
    f();
    return TQ_OK;
  } catch (const TantivyError &e) {
    g_err = e.what();
    switch (e.kind) {
      case TantivyError::InvalidArgument: return TQ_ERR_INVALID;
      case TantivyError::DataCorruption: return TQ_ERR_FORMAT;
      case TantivyError::Unsupported: return TQ_ERR_UNSUPPORTED;
      default: return TQ_ERR_HIP;
    }
  } catch (const std::exception &e) {
    g_err = e.what();
    return TQ_ERR_INVALID;
  }
}
}  // namespace

struct tqh_searcher {
  tq_ctx *ctx = nullptr;
  std::vector<std::shared_ptr<SegmentReader>> segments;
  std::unique_ptr<Searcher> searcher;
  std::vector<Weight> prepared;  // weights of the last tqh_prepare_batch
  std::vector<Weight> prepared_next;  // ... of the last tqh_prepare_batch_next: the batch AFTER the one being executed
};

extern "C" {

struct tqh_term_info {
  uint32_t term_id;
  uint32_t doc_freq;
  uint64_t postings_start, postings_end, positions_start, positions_end;
};
// mode: enum tq_mode (0 AND / all Must, 1 OR / all Should, 2 PHRASE, 3 BOOL = term clauses with
//       the given occurs, src/query/occur.rs order) + TQH_MODE_TERM (4) = TermQuery
#define TQH_MODE_TERM 4
struct tqh_query {
  uint8_t mode;
  uint32_t n_terms;
  const uint32_t *terms;
  const uint32_t *phrase_offsets;  // may be null
  const uint8_t *occurs;           // TQ_MODE_BOOL: 0 Should, 1 Must, 2 MustNot
  const uint8_t *clause_of;        // TQ_MODE_BOOL: terms sharing a value form one nested union; or null
  uint32_t min_should_match;       // TQ_MODE_BOOL
  const float *boosts;             // BoostQuery factor per term query (PHRASE: boosts[0] = the phrase's); or null
  const uint8_t *nested_occurs;    // TQ_MODE_BOOL, or null: see tantivy_amd_host.h
  const uint8_t *clause_min_should;  // TQ_MODE_BOOL, or null: minimum_number_should_match of the nested query, by clause_of value
  const uint8_t *atom_of;            // TQ_MODE_BOOL, or null: terms of one clause_of group sharing a value form a nested intersection
};

const char *tqh_last_error(void) { return g_err.c_str(); }

int tqh_searcher_new(tq_ctx *ctx, tqh_searcher **out) {
  return guard([&] {
    if (!ctx || !out) throw TantivyError(TantivyError::InvalidArgument, "null argument");
    auto *s = new tqh_searcher();
    s->ctx = ctx;
    *out = s;
  });
}
void tqh_searcher_free(tqh_searcher *s) { delete s; }

// The doc matrix's columns go to the densest lists of the segment (every TermInfo is known when a
// segment is added: from the caller's array or from its TermInfoStore), not to whichever dense lists the
// first queries happen to touch — the union kernels' time depended on the order of the first queries.
static void reserve_densest_columns(SegmentReader &seg, std::vector<std::pair<uint32_t, uint64_t>> &by_df) {
  const size_t keep = std::min<size_t>(by_df.size(), 40);  // (TQD_MAT_SLOTS columns)
  std::partial_sort(by_df.begin(), by_df.begin() + keep, by_df.end(),
                    [](const std::pair<uint32_t, uint64_t> &a, const std::pair<uint32_t, uint64_t> &b) {
                      return a.first != b.first ? a.first > b.first : a.second < b.second;
                    });
  std::vector<uint64_t> offs;
  for (size_t i = 0; i < keep; ++i) offs.push_back(by_df[i].second);
  if (tq_segment_reserve_columns(seg.raw(), offs.data(), (uint32_t)offs.size()) != TQ_OK)
    throw TantivyError(TantivyError::SystemError, tq_last_error());
}
static void reserve_densest_columns(SegmentReader &seg, const TermInfoStore &st) {
  std::vector<std::pair<uint32_t, uint64_t>> by_df;
  by_df.reserve(st.num_terms());
  for (uint64_t o = 0; o < st.num_terms(); ++o) {
    const TermInfo ti = st.get(o);
    by_df.emplace_back(ti.doc_freq, ti.postings_start);
  }
  reserve_densest_columns(seg, by_df);
}

int tqh_searcher_add_segment(tqh_searcher *s, int device, uint32_t max_doc, uint8_t record_option,
                             const uint8_t *idx, size_t idx_len, const uint8_t *pos,
                             size_t pos_len, const uint8_t *fieldnorm, size_t fn_len,
                             const tqh_term_info *terms, uint32_t n_terms) {
  return guard([&] {
    if (!s) throw TantivyError(TantivyError::InvalidArgument, "null searcher");
    auto seg = std::make_shared<SegmentReader>(s->ctx, device, (uint32_t)s->segments.size(),
                                               max_doc, record_option, idx, idx_len, pos, pos_len,
                                               fieldnorm, fn_len);
    for (uint32_t i = 0; i < n_terms; ++i) {
      TermInfo ti;
      ti.doc_freq = terms[i].doc_freq;
      ti.postings_start = terms[i].postings_start;
      ti.postings_end = terms[i].postings_end;
      ti.positions_start = terms[i].positions_start;
      ti.positions_end = terms[i].positions_end;
      seg->add_term(terms[i].term_id, ti);
    }
    {
      std::vector<std::pair<uint32_t, uint64_t>> by_df;
      by_df.reserve(n_terms);
      for (uint32_t i = 0; i < n_terms; ++i) by_df.emplace_back(terms[i].doc_freq, terms[i].postings_start);
      reserve_densest_columns(*seg, by_df);
    }
    s->segments.push_back(seg);
    s->searcher.reset(new Searcher(s->segments));
  });
}

// A segment whose term ids are the term ordinals of its TermInfoStore (the value half of the
// field's term dictionary file): no per-term TermInfo crosses the boundary.
int tqh_searcher_add_segment_with_store(tqh_searcher *s, int device, uint32_t max_doc,
                                        uint8_t record_option, const uint8_t *idx, size_t idx_len,
                                        const uint8_t *pos, size_t pos_len,
                                        const uint8_t *fieldnorm, size_t fn_len,
                                        const uint8_t *store, size_t store_len) {
  return guard([&] {
    if (!s) throw TantivyError(TantivyError::InvalidArgument, "null searcher");
    auto st = std::make_shared<TermInfoStore>(TermInfoStore::open(store, store_len));
    auto seg = std::make_shared<SegmentReader>(s->ctx, device, (uint32_t)s->segments.size(),
                                               max_doc, record_option, idx, idx_len, pos, pos_len,
                                               fieldnorm, fn_len);
    seg->set_term_info_store(st);
    reserve_densest_columns(*seg, *st);
    s->segments.push_back(seg);
    s->searcher.reset(new Searcher(s->segments));
  });
}

// The same for a segment whose sub-files are device-resident (tq_segment_upload_device).
int tqh_searcher_add_segment_device_with_store(tqh_searcher *s, int device, uint32_t max_doc,
                                               uint8_t record_option, const uint8_t *d_idx,
                                               size_t idx_len, const uint8_t *d_pos, size_t pos_len,
                                               const uint8_t *d_fieldnorm, size_t fn_len,
                                               uint64_t total_num_tokens, const uint8_t *store,
                                               size_t store_len) {
  return guard([&] {
    if (!s) throw TantivyError(TantivyError::InvalidArgument, "null searcher");
    auto st = std::make_shared<TermInfoStore>(TermInfoStore::open(store, store_len));
    auto seg = std::make_shared<SegmentReader>(SegmentReader::DeviceResident{}, s->ctx, device,
                                               (uint32_t)s->segments.size(), max_doc, record_option,
                                               d_idx, idx_len, d_pos, pos_len, d_fieldnorm, fn_len,
                                               total_num_tokens);
    seg->set_term_info_store(st);
    reserve_densest_columns(*seg, *st);
    s->segments.push_back(seg);
    s->searcher.reset(new Searcher(s->segments));
  });
}

// Bm25 statistics of a segment that lives on another rank (one segment per GPU).
int tqh_searcher_add_remote_stats(tqh_searcher *s, uint64_t max_doc, uint64_t total_num_tokens,
                                  const uint32_t *term_ids, const uint32_t *doc_freqs,
                                  uint32_t n_terms) {
  return guard([&] {
    if (!s || !s->searcher) throw TantivyError(TantivyError::InvalidArgument, "no segments");
    std::vector<std::pair<uint32_t, uint32_t>> v;
    for (uint32_t i = 0; i < n_terms; ++i) v.emplace_back(term_ids[i], doc_freqs[i]);
    s->searcher->add_remote_statistics(max_doc, total_num_tokens, v);
  });
}

// tqh_query -> Query (the shapes of tantivy_amd_host.h)
static Query build_query(const tqh_query &q) {
  Query query;
  if (q.mode > TQH_MODE_TERM)
    throw TantivyError(TantivyError::InvalidArgument, "unknown query mode");
  if (q.n_terms == 0) throw TantivyError(TantivyError::InvalidArgument, "query without terms");
  if (q.mode == TQ_MODE_BOOL) {
    if (!q.occurs) throw TantivyError(TantivyError::InvalidArgument, "TQ_MODE_BOOL needs occurs");
    std::vector<std::pair<Occur, Query>> clauses;
    std::vector<int> ids;  // clause_of value of every clause built so far
    std::vector<uint32_t> first_term_of;  // index (in q.terms) of the clause's first term
    std::vector<std::vector<int>> atom_ids(TQ_MAX_TERMS + 1);  // per clause: atom_of value of every member built so far
    std::vector<char> nested_built(TQ_MAX_TERMS + 1, 0);       // per clause: already a nested BooleanQuery
    // a term of a PhraseQuery inside the boolean query: nested_occurs | TQ_NESTED_PHRASE, its offset in phrase_offsets
    auto in_phrase = [&](uint32_t tt) {
      return q.nested_occurs && q.nested_occurs[tt] != 255 && (q.nested_occurs[tt] & TQ_NESTED_PHRASE) != 0;
    };
    auto leaf_of = [&](uint32_t tt) {
      if (in_phrase(tt))  // (the phrase's first term: the others are appended to phrase_terms below)
        return Query::phrase_with_offsets({{q.phrase_offsets ? q.phrase_offsets[tt] : 0u, q.terms[tt]}}).boosted(q.boosts ? q.boosts[tt] : 1.0f);
      return Query::term_query(q.terms[tt]).boosted(q.boosts ? q.boosts[tt] : 1.0f);
    };
    for (uint32_t t = 0; t < q.n_terms; ++t) {
      if (q.occurs[t] > 2) throw TantivyError(TantivyError::InvalidArgument, "bad occur");
      const Occur oc = q.occurs[t] == 1 ? Occur::Must
                                        : (q.occurs[t] == 2 ? Occur::MustNot : Occur::Should);
      const int id = q.clause_of ? (int)q.clause_of[t] : -1;
      size_t c = ids.size();
      if (id >= 0)
        for (c = 0; c < ids.size() && ids[c] != id; ++c) {}
      if (c == ids.size()) {
        ids.push_back(id);
        first_term_of.push_back(t);
        clauses.emplace_back(oc, leaf_of(t));
        continue;
      }
      if (clauses[c].first != oc)
        throw TantivyError(TantivyError::InvalidArgument, "a clause mixes occurs");
      Query &sub = clauses[c].second;
      // occur of a term INSIDE its clause: Should (a nested union) unless nested_occurs says
      // otherwise (`+a +(+b -c)`: clause_of {0,1,1}, occurs {1,1,1}, nested_occurs {255,1,2})
      auto inner = [&](uint32_t tt) {
        uint8_t v = q.nested_occurs ? q.nested_occurs[tt] : 255;
        if (v != 255) v &= (uint8_t)~(TQ_NESTED_PHRASE | TQ_NESTED_ANY);
        if (v != 255 && v > 2) throw TantivyError(TantivyError::InvalidArgument, "bad nested occur");
        return v == 1 ? Occur::Must : (v == 2 ? Occur::MustNot : Occur::Should);
      };
      if (!nested_built[c]) {  // second term of the clause: it becomes a nested query
        const Query first = sub;
        sub = Query::boolean({{inner(first_term_of[c]), first}});
        atom_ids[c] = {q.atom_of ? (int)q.atom_of[first_term_of[c]] : -1};
        nested_built[c] = true;
      }
      // atom_of: terms of the clause sharing a value are one member — an intersection of terms one level down
      // (`+a +((+b +c) d)`) or a phrase, with the member's occur = nested_occurs of its terms
      const int aid = q.atom_of ? (int)q.atom_of[t] : -1;
      size_t m = sub.clauses.size();
      if (aid >= 0)
        for (m = 0; m < atom_ids[c].size() && atom_ids[c][m] != aid; ++m) {}
      if (m >= sub.clauses.size()) {
        sub.clauses.emplace_back(inner(t), leaf_of(t));
        atom_ids[c].push_back(aid);
        continue;
      }
      if (sub.clauses[m].first != inner(t))
        throw TantivyError(TantivyError::InvalidArgument, "a conjunction mixes nested occurs");
      Query &member = sub.clauses[m].second;
      if (member.kind == Query::Phrase || in_phrase(t)) {
        if (member.kind != Query::Phrase || !in_phrase(t))
          throw TantivyError(TantivyError::InvalidArgument, "a member mixes phrase and plain terms");
        member.phrase_terms.emplace_back(q.phrase_offsets ? q.phrase_offsets[t] : (uint32_t)member.phrase_terms.size(), q.terms[t]);
        continue;
      }
      // (TQ_NESTED_ANY: the member is a UNION of its terms — a BooleanQuery of Should terms one level down)
      const bool any = q.nested_occurs && q.nested_occurs[t] != 255 && (q.nested_occurs[t] & TQ_NESTED_ANY) != 0;
      const Occur leaf_occur = any ? Occur::Should : Occur::Must;
      if (member.kind == Query::Term) member = Query::boolean({{leaf_occur, Query(member)}});
      if (!member.clauses.empty() && member.clauses[0].first != leaf_occur)
        throw TantivyError(TantivyError::InvalidArgument, "a member mixes union and intersection terms");
      member.clauses.emplace_back(leaf_occur, leaf_of(t));
    }
    // a clause that is ONE phrase (`+"a b"`): the one-member nested query is the phrase itself
    for (size_t c = 0; c < clauses.size(); ++c) {
      Query &sub = clauses[c].second;
      if (nested_built[c] && sub.clauses.size() == 1 && sub.clauses[0].second.kind == Query::Phrase &&
          sub.clauses[0].first != Occur::MustNot && !(q.clause_min_should && ids[c] >= 0 && ids[c] < (int)TQ_MAX_TERMS && q.clause_min_should[ids[c]])) {
        Query ph = sub.clauses[0].second;
        sub = std::move(ph);
      }
    }
    if (q.clause_min_should)  // BooleanQuery::set_minimum_number_should_match on the nested queries
      for (size_t c = 0; c < clauses.size(); ++c)
        if (ids[c] >= 0 && ids[c] < (int)TQ_MAX_TERMS && q.clause_min_should[ids[c]]) {
          Query &sub = clauses[c].second;
          if (sub.kind == Query::Term) sub = Query::boolean({{Occur::Should, sub}});  // (a one-term nested query)
          sub.set_minimum_number_should_match(q.clause_min_should[ids[c]]);
        }
    query = Query::boolean(std::move(clauses));
    query.set_minimum_number_should_match(q.min_should_match);
  } else if (q.mode == TQH_MODE_TERM || (q.n_terms == 1 && q.mode != TQ_MODE_PHRASE)) {
    query = Query::term_query(q.terms[0]).boosted(q.boosts ? q.boosts[0] : 1.0f);
  } else if (q.mode == TQ_MODE_PHRASE) {
    std::vector<std::pair<uint32_t, uint32_t>> pt;
    for (uint32_t t = 0; t < q.n_terms; ++t)
      pt.emplace_back(q.phrase_offsets ? q.phrase_offsets[t] : t, q.terms[t]);
    query = Query::phrase_with_offsets(std::move(pt));
    // BoostQuery around the PhraseQuery: PhraseWeight applies it with boost_by
    // (phrase_weight.rs:42-69); boosts[0] is the factor of the whole phrase
    if (q.boosts) query.boosted(q.boosts[0]);
  } else {
    std::vector<std::pair<Occur, Query>> clauses;
    for (uint32_t t = 0; t < q.n_terms; ++t)
      clauses.emplace_back(q.mode == TQ_MODE_AND ? Occur::Must : Occur::Should,
                           Query::term_query(q.terms[t]).boosted(q.boosts ? q.boosts[t] : 1.0f));
    query = Query::boolean(std::move(clauses));
  }
  return query;
}

// Query::weight for a batch (global statistics, executor choice).  Kept so that a benchmark can
// time execution separately from weight construction, like tantivy's own benches do.
static void prepare_into(tqh_searcher *s, const tqh_query *queries, uint32_t n, std::vector<Weight> &out) {
  if (!s || !s->searcher) throw TantivyError(TantivyError::InvalidArgument, "no segments");
  static const bool trace = getenv("TQH_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  struct Report {
    bool on;
    std::chrono::steady_clock::time_point t0;
    uint32_t n;
    ~Report() {
      if (on)
        fprintf(stderr, "[tqh] prepare %u queries: %.3f ms\n", n,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
  } report{trace, t0, n};
  out.clear();
  out.resize(n);
  Searcher::FlatContext fc;  // (statistics + the field's tf cache: looked up with the first flat query of the batch)
  for (uint32_t i = 0; i < n; ++i) {
    const tqh_query &q = queries[i];
    // unboosted all-Must / all-Should term clauses (what a query parser makes of `+a +b` / `a b c`): the weight
    // without the detour through a Query tree — the same f32 arithmetic as Searcher::weight
    if ((q.mode == TQ_MODE_AND || q.mode == TQ_MODE_OR) && !q.boosts && q.n_terms >= 1 && q.terms) {
      if (!fc.cache) fc = s->searcher->flat_context();
      s->searcher->weight_flat_into(fc, q.mode, q.terms, q.n_terms, out[i]);
    } else {
      out[i] = s->searcher->weight(build_query(q));
    }
  }
}
int tqh_prepare_batch(tqh_searcher *s, const tqh_query *queries, uint32_t n) {
  return guard([&] { prepare_into(s, queries, n, s->prepared); });
}
// The serving pattern's pipeline: Query::weight of batch i + 1 on one thread while another executes batch i
// (Searcher::weight is thread-safe: concurrent Searcher::search calls do the same).  tqh_prepare_batch_next fills a
// second slot and touches nothing the execution of the current batch reads; tqh_commit_next makes that slot the
// current batch (call it from the executing thread, between two batches, after the preparing call has returned).
int tqh_prepare_batch_next(tqh_searcher *s, const tqh_query *queries, uint32_t n) {
  return guard([&] { prepare_into(s, queries, n, s->prepared_next); });
}
int tqh_commit_next(tqh_searcher *s) {
  return guard([&] {
    if (!s) throw TantivyError(TantivyError::InvalidArgument, "null searcher");
    s->prepared.swap(s->prepared_next);
  });
}

// Searcher::search for the prepared batch with TopDocs::with_limit(limit).and_offset(offset).
int tqh_search_prepared(tqh_searcher *s, uint32_t offset, uint32_t limit, float *scores,
                        uint32_t *segment_ords, uint32_t *docs, uint32_t *counts) {
  return guard([&] {
    if (!s || !s->searcher) throw TantivyError(TantivyError::InvalidArgument, "no segments");
    const TopDocs td = TopDocs::with_limit(limit).and_offset(offset);
    std::vector<Fruit> res = s->searcher->search_batch(s->prepared, td);
    for (size_t q = 0; q < res.size(); ++q) {
      counts[q] = (uint32_t)res[q].size();
      for (uint32_t i = 0; i < limit; ++i) {
        const bool real = i < res[q].size();
        scores[q * limit + i] = real ? res[q][i].first : 0.0f;
        segment_ords[q * limit + i] = real ? res[q][i].second.segment_ord : 0xFFFFFFFFu;
        docs[q * limit + i] = real ? res[q][i].second.doc_id : TERMINATED;
      }
    }
  });
}

// Searcher::search called from n_threads host threads at once, one query per call (the reference's
// own call pattern, searcher.rs:180-238): thread t takes queries t, t + n_threads, ...  Every call
// is Query::weight + one collect_segment per segment (tq_search_one: the calls of concurrent
// threads ride in shared launches) + merge_fruits.  Outputs [n][limit] as tqh_search_prepared;
// latency_ms[q] = wall time of query q's call (or null); *wall_ms = the whole run.
int tqh_search_concurrent(tqh_searcher *s, const tqh_query *queries, uint32_t n, uint32_t offset,
                          uint32_t limit, uint32_t n_threads, float *scores, uint32_t *segment_ords,
                          uint32_t *docs, uint32_t *counts, float *latency_ms, double *wall_ms) {
  return guard([&] {
    if (!s || !s->searcher) throw TantivyError(TantivyError::InvalidArgument, "no segments");
    if (!queries || !scores || !segment_ords || !docs || !counts || !n_threads)
      throw TantivyError(TantivyError::InvalidArgument, "null argument");
    const TopDocs td = TopDocs::with_limit(limit).and_offset(offset);
    std::vector<std::string> errors(n_threads);
    std::vector<TantivyError::Kind> kinds(n_threads, TantivyError::InvalidArgument);
    // (a server's request threads exist before the requests do: the clock starts when every thread stands at the
    // line — creating 1 024 threads took a fifth of the wall time of a 10 000-query run)
    static const bool conc_trace = getenv("TQH_CONC_TRACE") != nullptr;  // (tools/latency_threads.py: the callers' own CPU time)
    std::atomic<uint64_t> callers_cpu_ns{0};
    // (asleep, not spinning: a thousand threads yielding in a loop burn the process's CPU quota before the first query)
    std::mutex line_m;
    std::condition_variable line_cv;
    uint32_t at_line = 0;
    bool go = false;
    std::chrono::steady_clock::time_point t0;
    auto work = [&](uint32_t t) {
      {
        std::unique_lock<std::mutex> ll(line_m);
        if (++at_line == n_threads) {
          t0 = std::chrono::steady_clock::now();
          go = true;
          line_cv.notify_all();
        } else {
          line_cv.wait(ll, [&] { return go; });
        }
      }
      timespec c0{}, c1{};
      if (conc_trace) clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
      try {
        for (uint32_t qi = t; qi < n; qi += n_threads) {
          const auto q0 = std::chrono::steady_clock::now();
          const Fruit res = s->searcher->search(build_query(queries[qi]), td);
          if (latency_ms)
            latency_ms[qi] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - q0).count();
          counts[qi] = (uint32_t)res.size();
          for (uint32_t i = 0; i < limit; ++i) {
            const bool real = i < res.size();
            scores[(size_t)qi * limit + i] = real ? res[i].first : 0.0f;
            segment_ords[(size_t)qi * limit + i] = real ? res[i].second.segment_ord : 0xFFFFFFFFu;
            docs[(size_t)qi * limit + i] = real ? res[i].second.doc_id : TERMINATED;
          }
        }
      } catch (const TantivyError &e) {
        errors[t] = e.what();
        kinds[t] = e.kind;
      } catch (const std::exception &e) {
        errors[t] = e.what();
      }
      if (conc_trace) {
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
        callers_cpu_ns.fetch_add((uint64_t)((c1.tv_sec - c0.tv_sec) * 1000000000ll + (c1.tv_nsec - c0.tv_nsec)), std::memory_order_relaxed);
      }
    };
    std::vector<std::thread> threads;
    for (uint32_t t = 1; t < n_threads; ++t) threads.emplace_back(work, t);
    work(0);
    for (std::thread &th : threads) th.join();
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (conc_trace)
      fprintf(stderr, "[tqh] %u threads, %u queries: %.1f us of CPU per query inside the calling threads (from the starting line on)\n", n_threads, n,
              (double)callers_cpu_ns.load() / 1e3 / (double)std::max<uint32_t>(1u, n));
    for (uint32_t t = 0; t < n_threads; ++t)
      if (!errors[t].empty()) throw TantivyError(kinds[t], errors[t]);
  });
}

// Per-segment results of the prepared batch (collect_segment), for multi-process runs where
// each rank holds one segment and the cross-segment merge happens after an all-gather.
int tqh_collect_segment_prepared(tqh_searcher *s, uint32_t segment_ord, uint32_t k, float *scores,
                                 uint32_t *docs, uint32_t *counts) {
  return guard([&] {
    if (!s || !s->searcher || segment_ord >= s->segments.size())
      throw TantivyError(TantivyError::InvalidArgument, "bad segment");
    std::vector<float> sc;
    std::vector<uint32_t> dc, ct;
    s->searcher->collect_segment_batch(segment_ord, s->prepared, k, sc, dc, ct);
    std::memcpy(scores, sc.data(), sc.size() * sizeof(float));
    std::memcpy(docs, dc.data(), dc.size() * sizeof(uint32_t));
    std::memcpy(counts, ct.data(), ct.size() * sizeof(uint32_t));
  });
}

int tqh_collect_segment_prepared_device(tqh_searcher *s, uint32_t segment_ord, uint32_t k,
                                        float *d_scores, uint32_t *d_docs, uint32_t *d_counts,
                                        void *hip_stream) {
  return guard([&] {
    if (!s || !s->searcher || segment_ord >= s->segments.size())
      throw TantivyError(TantivyError::InvalidArgument, "bad segment");
    s->searcher->collect_segment_batch_device(segment_ord, s->prepared, k, d_scores, d_docs,
                                              d_counts, hip_stream);
  });
}

// Bm25Weight for tests / tools: weight and 256-entry cache from global statistics.
int tqh_bm25_for_terms(const uint64_t *term_doc_freqs, uint32_t n_terms, uint64_t total_num_docs,
                       uint64_t total_num_tokens, float boost, float *weight_out,
                       float *cache_out /*256 or null*/) {
  return guard([&] {
    if (!term_doc_freqs || !n_terms || !weight_out)
      throw TantivyError(TantivyError::InvalidArgument, "null argument");
    std::vector<uint64_t> dfs(term_doc_freqs, term_doc_freqs + n_terms);
    Bm25Weight w = Bm25Weight::for_terms(dfs, total_num_docs, total_num_tokens).boost_by(boost);
    *weight_out = w.weight;
    if (cache_out) std::memcpy(cache_out, w.cache, sizeof w.cache);
  });
}

tq_segment *tqh_segment_raw(tqh_searcher *s, uint32_t segment_ord) {
  if (!s || segment_ord >= s->segments.size()) return nullptr;
  return s->segments[segment_ord]->raw();
}
uint32_t tqh_term_handle(tqh_searcher *s, uint32_t segment_ord, uint32_t term_id) {
  if (!s || segment_ord >= s->segments.size()) return TQ_TERM_ABSENT;
  uint32_t h = TQ_TERM_ABSENT;
  guard([&] { h = s->segments[segment_ord]->term_handle(term_id); });
  return h;
}

// Searcher::search(&query, &Count) of the prepared batch: counts[q] = alive matching docs over
// all segments.
int tqh_count_prepared(tqh_searcher *s, uint64_t *counts) {
  return guard([&] {
    if (!s || !s->searcher || !counts) throw TantivyError(TantivyError::InvalidArgument, "null argument");
    const std::vector<uint64_t> c = s->searcher->count_batch(s->prepared);
    if (!c.empty()) std::memcpy(counts, c.data(), c.size() * sizeof(uint64_t));
  });
}

// ---- TermInfoStore (src/termdict/fst_termdict/term_info_store.rs)
struct tqh_term_info_store {
  TermInfoStore store;
};
int tqh_term_dictionary_values(const uint8_t *file, size_t len, uint64_t *store_off,
                               uint64_t *store_len) {
  return guard([&] {
    if (!store_off || !store_len) throw TantivyError(TantivyError::InvalidArgument, "null argument");
    size_t o = 0, l = 0;
    term_dictionary_values(file, len, &o, &l);
    *store_off = o;
    *store_len = l;
  });
}
int tqh_term_info_store_open(const uint8_t *bytes, size_t len, tqh_term_info_store **out) {
  return guard([&] {
    if (!out) throw TantivyError(TantivyError::InvalidArgument, "null argument");
    *out = new tqh_term_info_store{TermInfoStore::open(bytes, len)};
  });
}
void tqh_term_info_store_free(tqh_term_info_store *s) { delete s; }
uint64_t tqh_term_info_store_num_terms(const tqh_term_info_store *s) {
  return s ? s->store.num_terms() : 0;
}
int tqh_term_info_store_get(const tqh_term_info_store *s, const uint64_t *term_ords, uint32_t n,
                            tqh_term_info *out) {
  return guard([&] {
    if (!s || (n && (!term_ords || !out))) throw TantivyError(TantivyError::InvalidArgument, "null argument");
    for (uint32_t i = 0; i < n; ++i) {
      const TermInfo ti = s->store.get(term_ords[i]);
      out[i].term_id = (uint32_t)term_ords[i];
      out[i].doc_freq = ti.doc_freq;
      out[i].postings_start = ti.postings_start;
      out[i].postings_end = ti.postings_end;
      out[i].positions_start = ti.positions_start;
      out[i].positions_end = ti.positions_end;
    }
  });
}
// TermInfoStoreWriter over term infos in ordinal order; *out_len = bytes needed
int tqh_term_info_store_write(const tqh_term_info *infos, uint32_t n, uint8_t *out, uint64_t out_cap,
                              uint64_t *out_len) {
  return guard([&] {
    if (!out_len || (n && !infos)) throw TantivyError(TantivyError::InvalidArgument, "null argument");
    TermInfoStoreWriter w;
    for (uint32_t i = 0; i < n; ++i) {
      TermInfo ti;
      ti.doc_freq = infos[i].doc_freq;
      ti.postings_start = infos[i].postings_start;
      ti.postings_end = infos[i].postings_end;
      ti.positions_start = infos[i].positions_start;
      ti.positions_end = infos[i].positions_end;
      if (ti.postings_end < ti.postings_start || ti.positions_end < ti.positions_start)
        throw TantivyError(TantivyError::InvalidArgument, "term info range ends before it starts");
      w.write_term_info(ti);
    }
    std::vector<uint8_t> bytes;
    w.serialize(bytes);
    *out_len = bytes.size();
    if (bytes.size() > out_cap) throw TantivyError(TantivyError::InvalidArgument, "output buffer too small");
    if (!bytes.empty()) std::memcpy(out, bytes.data(), bytes.size());
  });
}

}  // extern "C"
