// tq_count_batch: the Count collector (src/collector/count_collector.rs:39-80).  Queries whose lists all
// have a bitmap and whose doc set is cheaper to get from bitmap words than from postings are counted by
// tq_count.hip (a bitwise expression per 32 docs); the others by an exhaustive scan with the smallest top-k.
#include "tq_internal.hpp"

#include <unordered_map>

namespace tqi {

// The query as a bitwise expression, or false if it has to be scanned (a phrase, a list without a bitmap,
// minimum_number_should_match >= 2 over fewer Should clauses than that, malformed input — the scan reports it).
// `known` = the count is known without looking (an absent Must term, MustNot clauses only, ...): 0 matches.
bool count_expression(tq_segment *s, const tq_query &q, TqkCountQuery &cq, bool &known, uint64_t &driver_postings,
                      std::unordered_map<uint32_t, uint32_t> &temp_slot, uint32_t max_temp) {
  known = false;
  driver_postings = 0;
  cq = TqkCountQuery{};
  if (!q.terms || q.n_terms == 0 || q.n_terms > TQ_MAX_TERMS || q.mode == TQ_MODE_PHRASE || q.mode > TQ_MODE_BOOL) return false;
  if (q.mode == TQ_MODE_BOOL && !q.occurs) return false;
  if (bool_query_is_tree(q)) return false;  // (nested queries: the scan — tq_tree.hip — counts them from bitmap words itself)
  struct Clause {
    uint32_t id, occur, n = 0, terms[TQ_MAX_TERMS];
    uint64_t cost = 0;
  };
  Clause cl[TQ_MAX_TERMS];
  uint32_t n_cl = 0;
  for (uint32_t i = 0; i < q.n_terms; ++i) {
    uint32_t occur = q.mode == TQ_MODE_AND ? TQ_MUST : TQ_SHOULD;
    uint32_t id = i;
    if (q.mode == TQ_MODE_BOOL) {
      occur = q.occurs[i];
      if (occur > TQ_MUST_NOT) return false;
      if (q.clause_of) id = q.clause_of[i];
    }
    uint32_t c = 0;
    while (c < n_cl && cl[c].id != id) ++c;
    if (c == n_cl) {
      cl[n_cl].id = id;
      cl[n_cl].occur = occur;
      ++n_cl;
    } else if (cl[c].occur != occur) {
      return false;  // (mixed occurs in one clause: the scan reports it)
    }
    const uint32_t h = q.terms[i];
    if (h == TQ_TERM_ABSENT) continue;
    if (h >= s->terms.size()) return false;
    if (!(s->terms[h].dense_blob && s->opt.use_dense)) {
      // a list without a bitmap gets one for the duration of the batch (count_scatter_kernel), while the
      // batch's scratch has room for another
      if (!temp_slot.count(h) && temp_slot.size() >= max_temp) return false;
    }
    cl[c].terms[cl[c].n++] = h;
    cl[c].cost += s->terms[h].doc_freq;
  }
  // BooleanWeight::complex_scorer (boolean_weight.rs:236-431), as plan_bool_query restates it
  uint32_t n_must = 0, n_should = 0;
  bool empty = false;
  for (uint32_t c = 0; c < n_cl; ++c) {
    if (cl[c].occur == TQ_MUST) {
      if (cl[c].n == 0) empty = true;
      ++n_must;
    } else if (cl[c].n) {
      if (cl[c].occur == TQ_SHOULD) ++n_should;
    }
  }
  uint32_t msm = q.mode == TQ_MODE_BOOL ? q.min_should_match : 0u;
  if (msm > n_should) empty = true;
  bool should_is_must = false;
  if (!empty && msm >= 2) {
    if (msm != n_should) return false;  // "at least m of n": not a bitwise expression of this shape
    should_is_must = true;              // all of them: Must clauses
    msm = 0;
  }
  if (n_must == 0 && n_should == 0) empty = true;
  if (empty) {
    known = true;
    return true;
  }
  uint32_t n = 0;
  uint64_t must_cost = ~0ull, should_cost = 0;
  auto put = [&](const Clause &c, uint32_t kind, bool clause_union) {
    for (uint32_t i = 0; i < c.n; ++i) {
      const TermHost &th = s->terms[c.terms[i]];
      if (th.dense_blob && s->opt.use_dense) {
        cq.dense[n] = (const uint2 *)th.dense_blob;
      } else {  // (the slot for now; the pointer once the scratch is allocated)
        const uint32_t slot = temp_slot.emplace(c.terms[i], (uint32_t)temp_slot.size()).first->second;
        cq.dense[n] = (const uint2 *)(uintptr_t)slot;
        cq.narrow |= 1u << n;
      }
      cq.kinds |= kind << (2u * n);
      if (kind == TQK_COUNT_MUST && (!clause_union || i + 1 == c.n)) cq.clause_end |= 1u << n;
      ++n;
    }
  };
  const bool has_must = n_must > 0 || should_is_must;
  for (uint32_t c = 0; c < n_cl; ++c) {
    if (cl[c].occur == TQ_MUST || (should_is_must && cl[c].occur == TQ_SHOULD && cl[c].n)) {
      put(cl[c], TQK_COUNT_MUST, true);
      must_cost = std::min(must_cost, cl[c].cost);
    }
  }
  for (uint32_t c = 0; c < n_cl; ++c)
    if (cl[c].occur == TQ_MUST_NOT && cl[c].n) put(cl[c], TQK_COUNT_NOT, false);
  if (!should_is_must)
    for (uint32_t c = 0; c < n_cl; ++c)
      if (cl[c].occur == TQ_SHOULD && cl[c].n) {
        // with Must clauses and no minimum the Should lists do not change the doc set
        if (has_must && msm == 0) continue;
        put(cl[c], TQK_COUNT_SHOULD, false);
        should_cost += cl[c].cost;
      }
  cq.n_terms = n;
  cq.flags = (has_must ? TQK_COUNT_HAS_MUST : 0u) | ((has_must && msm == 1) ? TQK_COUNT_NEED_SHOULD : 0u);
  driver_postings = has_must ? must_cost : should_cost;  // what a scan would walk: the cheapest Must clause / the union
  return true;
}

int count_batch(tq_segment *s, const tq_query *queries, uint32_t n_queries, uint32_t *out_counts) {
  // A bitmap word costs 8 bytes per list per 32 docs whatever the lists hold; a scan decodes the
  // postings of the driving clause (and probes the others): bitmaps when the driving clause holds at
  // least max_doc / ratio postings per list of the expression ("count_bitmap_ratio", 0 = never).
  static const uint32_t kRatioEnv = tune_u32("TQ_COUNT_BITMAP_RATIO", 0xFFFFFFFFu);
  const uint32_t kRatio = kRatioEnv != 0xFFFFFFFFu ? kRatioEnv : (uint32_t)s->opt.count_bitmap_ratio;
  static const uint64_t kTempBudget = (uint64_t)std::max<uint32_t>(1u, tune_u32("TQ_COUNT_TEMP_MB", 1024)) << 20;
  const uint32_t n_words = (uint32_t)(((uint64_t)s->max_doc + 31u) / 32u);
  const uint32_t words_per_list = (n_words + 63u) & ~63u;
  const uint32_t max_temp = (uint32_t)std::min<uint64_t>(4096u, kTempBudget / ((uint64_t)words_per_list * 4u));
  std::vector<TqkCountQuery> cqs;
  std::vector<uint32_t> bitmap_q, scan_q;
  std::unordered_map<uint32_t, uint32_t> temp_slot, trial;  // lists without a bitmap -> slot of the batch's scratch
  for (uint32_t qi = 0; qi < n_queries; ++qi) {
    TqkCountQuery cq;
    bool known = false;
    uint64_t driver = 0;
    trial = temp_slot;  // (a query that ends up scanned leaves no slots behind)
    const bool expr = kRatio && count_expression(s, queries[qi], cq, known, driver, trial, max_temp);
    if (expr && known) {
      out_counts[qi] = 0;
    } else if (expr && driver * kRatio >= (uint64_t)cq.n_terms * s->max_doc) {
      cqs.push_back(cq);
      bitmap_q.push_back(qi);
      temp_slot.swap(trial);
    } else {
      scan_q.push_back(qi);
    }
  }
  HIP_TRY(hipSetDevice(s->device));
  uint32_t mask = 0;
  uint64_t algo_bytes = 0;
  if (!scan_q.empty()) {
    // every match has to be visited: exhaustive scan, smallest top-k
    const uint32_t n = (uint32_t)scan_q.size();
    std::vector<tq_query> qs(n);
    for (uint32_t i = 0; i < n; ++i) {
      qs[i] = queries[scan_q[i]];
      qs[i].k = 1;
    }
    std::vector<float> sc(n);
    std::vector<uint32_t> dc(n), ct(n), mc(n);
    CallOpts co;
    int rc = resolve_opts(s, nullptr, co);
    if (rc != TQ_OK) return rc;
    co.exhaustive = true;  // per call: the segment's options are not touched
    rc = search_batch_host(s, qs.data(), n, 1, sc.data(), dc.data(), ct.data(), co);
    if (rc != TQ_OK) return rc;
    rc = tq_last_batch_match_counts(s, mc.data(), n);
    if (rc != TQ_OK) return rc;
    for (uint32_t i = 0; i < n; ++i) out_counts[scan_q[i]] = mc[i];
    mask = s->stats.kernel_mask;
    algo_bytes = s->stats.algorithmic_bytes;
  } else {
    const int wrc = wait_segment_idle(s);
    if (wrc != TQ_OK) return wrc;
  }
  if (!bitmap_q.empty()) {
    const uint32_t n = (uint32_t)bitmap_q.size();
    int rc = s->d_count_queries.ensure((size_t)n * sizeof(TqkCountQuery));
    if (rc == TQ_OK) rc = s->d_count_out.ensure((size_t)n * sizeof(uint32_t));
    if (rc != TQ_OK) return rc;
    if (!temp_slot.empty()) {  // the batch's lists without a bitmap, as bits
      const int src = sync_terms(s, s->stream);
      if (src != TQ_OK) return src;
      std::vector<uint4> wgs;
      for (const auto &kv : temp_slot)
        for (uint32_t j = 0; j < s->terms[kv.first].n_blocks; j += 4u) wgs.push_back(make_uint4(kv.first, j, kv.second, 0u));
      rc = s->d_count_bits.ensure((size_t)temp_slot.size() * words_per_list * sizeof(uint32_t));
      if (rc == TQ_OK) rc = s->d_count_wgs.ensure(wgs.size() * sizeof(uint4));
      if (rc != TQ_OK) return rc;
      HIP_TRY(hipMemsetAsync(s->d_count_bits.p, 0, (size_t)temp_slot.size() * words_per_list * sizeof(uint32_t), s->stream));
      HIP_TRY(hipMemcpyAsync(s->d_count_wgs.p, wgs.data(), wgs.size() * sizeof(uint4), hipMemcpyHostToDevice, s->stream));
      const hipError_t se = tqk_launch_count_scatter(s->dseg, s->d_terms, (const uint4 *)s->d_count_wgs.p, (uint32_t)wgs.size(),
                                                     (uint32_t *)s->d_count_bits.p, words_per_list, s->stream);
      if (se != hipSuccess) return fail(TQ_ERR_HIP, "count scatter launch: %s", hipGetErrorString(se));
      for (TqkCountQuery &cq : cqs)
        for (uint32_t m = 0; m < cq.n_terms; ++m)
          if ((cq.narrow >> m) & 1u)
            cq.dense[m] = (const uint2 *)((const uint32_t *)s->d_count_bits.p + (size_t)(uintptr_t)cq.dense[m] * words_per_list);
    }
    HIP_TRY(hipMemcpyAsync(s->d_count_queries.p, cqs.data(), (size_t)n * sizeof(TqkCountQuery), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_count_out.p, 0, (size_t)n * sizeof(uint32_t), s->stream));
    TqkCountParams p{};
    p.queries = (const TqkCountQuery *)s->d_count_queries.p;
    p.alive = s->d_alive;
    p.out_counts = (uint32_t *)s->d_count_out.p;
    p.n_queries = n;
    p.n_words = n_words;
    const hipError_t e = tqk_launch_count_bitmaps(p, s->stream);
    if (e != hipSuccess) return fail(TQ_ERR_HIP, "count kernel launch: %s", hipGetErrorString(e));
    std::vector<uint32_t> got(n);
    HIP_TRY(hipMemcpyAsync(got.data(), s->d_count_out.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (uint32_t i = 0; i < n; ++i) {
      out_counts[bitmap_q[i]] = got[i];
      algo_bytes += (uint64_t)cqs[i].n_terms * p.n_words * 4u;  // the lists' bits (the rank halves of the words ride along)
    }
    mask |= TQ_KERNEL_COUNT_BITMAPS;
  }
  s->stats.kernel_mask = mask;
  s->stats.algorithmic_bytes = algo_bytes;
  return TQ_OK;
}

}  // namespace tqi
