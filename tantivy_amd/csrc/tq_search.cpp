// tq_search.cpp — one batch of queries on one segment: validation, launch groups, the planners, the staging blob,
// kernel launches and merges (tq_search_batch, tq_search_batch_device and their _opts forms)
// Part of the C ABI library of include/tantivy_amd.h (internal declarations: tq_internal.hpp).
#include "tq_internal.hpp"

namespace tqi {

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
  prep_apply_lmax(s, false);
  if (s->prep_pending >= 0) {  // terms uploaded by tq_term_prepare_batch on the copy stream: this batch reads them
    if (st != s->stream) HIP_TRY(hipStreamWaitEvent(st, s->ev_prep[s->prep_pending], 0));
    s->prep_pending = -1;
  }
  // Terms prepared since the last batch: their records are APPENDED to the device table through this batch's
  // staging blob (a device-to-device copy on the batch's stream, below) — the batches in flight never read beyond
  // the records they were planned with, so nothing waits for them.  A grown table, or a changed record that a
  // batch in flight may read, takes the blocking road (sync_terms).
  int rc = TQ_OK;
  size_t terms_from = 0, terms_to = 0;
  if (s->d_terms_dirty) {
    const size_t n_terms_now = s->h_dterms.size();
    if (s->d_terms && n_terms_now <= s->d_terms_cap && s->d_terms_dirty_from >= s->d_terms_synced) {
      terms_from = std::min(s->d_terms_dirty_from, n_terms_now);
      terms_to = n_terms_now;
    } else {
      rc = sync_terms(s, st);
      if (rc != TQ_OK) return rc;
    }
  }
  const int opt_exhaustive = co.exhaustive ? 1 : 0;
  // Boolean queries ride in the shared leader-major launch if every list they probe has a bitmap + tf
  // bytes: lists below "dense_ratio" get them the first time a boolean query names them ("probe_budget_x")
  static const bool kUseBShare = tune_u32("TQ_BSHARE", 1) != 0;
  static const bool kBsRdir = tune_u32("TQ_BS_RDIR", 1) != 0;  // ... or a range directory (0: probe tables for those too)
  // 2-term intersections that probe a list without tables of its own: on the per-query kernel they run on
  // the side stream next to the shared launch, almost for free while they are few (5 % of the headline
  // batch: pulling them into the shared launch cost 7 % of kernel time — no doc-matrix column, so nothing
  // filters their docs before the scoring stage — for the same step time); when they are many (22 % at
  // 4 096 terms) the shared launch with probe tables is 12 % faster.  TQ_AS_PROBE: 0 never, 1 from 10 % of
  // the batch (default), 2 always.
  static const uint32_t kProbeAnd = tune_u32("TQ_AS_PROBE", 1);
  // The address span of the side tables the shared launches reach through 32-bit offsets (8-byte units from its lower
  // end: 32 GB), taken again when tables were added.
  auto update_table_span = [&]() {
    if (s->share_span_terms != s->terms.size()) {  // (terms are prepared rarely)
      uint64_t lo = ~0ull, hi = 0;
      for (const TermHost &th : s->terms)
        for (const void *ptr : {th.dense_blob, th.tf8_blob, th.probe_dense_blob, th.probe_tf8_blob, th.rmax_blob, th.posdir_blob, th.probe_posdir_blob})
          if (ptr) {
            lo = std::min<uint64_t>(lo, (uint64_t)ptr);
            hi = std::max<uint64_t>(hi, (uint64_t)ptr);
          }
      if (lo == ~0ull) lo = hi = 8;
      {  // the range directories' chunks: inside the span when it still fits with them (else the lists do without)
        uint64_t lo2 = lo, hi2 = hi;
        for (const auto &ch : s->rdir_chunks) {
          lo2 = std::min<uint64_t>(lo2, (uint64_t)ch.first);
          hi2 = std::max<uint64_t>(hi2, (uint64_t)ch.first + ch.second);
        }
        s->rdir_span_ok = !s->rdir_chunks.empty() && hi2 - (lo2 - 8) < (8ull << 32);
        if (s->rdir_span_ok) {
          lo = lo2;
          hi = hi2;
        }
      }
      s->share_table_lo = (lo - 8) & ~255ull;  // (256-byte aligned: a range directory's offset keeps its low five bits free for the shift)
      s->share_span_ok = hi - s->share_table_lo < (8ull << 32);
      s->share_span_terms = s->terms.size();
    }
  };
  update_table_span();
  // ONE pass over the caller's queries for everything that only needs a look at them: which Bm25Weight cache each
  // uses (pointer identity; a handful per batch), how many are unions (a one-list intersection runs as one), whether
  // there are boolean queries at all, how many 2-term intersections probe a list without tables (six passes over
  // 10 000 80-byte descriptors were 0.16 ms of the headline step's 0.9 ms of host time)
  if (!s->plan) s->plan = new PlanScratch();
  PlanScratch &ps_plan = *s->plan;
  std::vector<const float *> caches;
  ps_plan.q_cache.resize(n_queries);
  uint32_t n_union_queries = 0, n_bool_queries = 0, n_sparse2 = 0;
  {
    const size_t n_terms_known = s->terms.size();
    const float *last_tc = nullptr;
    uint32_t last_idx = 0;
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const tq_query &q = queries[qi];
      const float *tc = q.tf_cache;
      uint32_t cache_idx = 0;
      if (tc == last_tc && tc) {
        cache_idx = last_idx;
      } else if (tc) {
        for (; cache_idx < caches.size(); ++cache_idx)
          if (caches[cache_idx] == tc) break;
        if (cache_idx == caches.size()) caches.push_back(tc);
        last_tc = tc;
        last_idx = cache_idx;
      }
      ps_plan.q_cache[qi] = cache_idx;
      n_union_queries += (q.mode == TQ_MODE_OR || (q.mode == TQ_MODE_AND && q.n_terms == 1)) ? 1u : 0u;
      n_bool_queries += q.mode == TQ_MODE_BOOL ? 1u : 0u;
      if (q.mode == TQ_MODE_AND && q.n_terms == 2 && q.terms && q.terms[0] < n_terms_known && q.terms[1] < n_terms_known) {
        const TermHost &a = s->terms[q.terms[0]], &b = s->terms[q.terms[1]];
        const TermHost &probed = b.doc_freq < a.doc_freq ? a : b;
        if (!(probed.dense_blob && probed.tf8_blob) && !(probed.rdir_blob && s->rdir_span_ok)) ++n_sparse2;
      }
    }
  }
  bool and_probe = false;
  if (kProbeAnd && !opt_exhaustive && s->opt.use_dense && s->opt.dense && s->opt.probe_budget_x > 0 &&
      n_queries >= (uint32_t)s->opt.ashare_min_batch)
    and_probe = kProbeAnd >= 2 || (uint64_t)n_sparse2 * 10u >= n_queries;
  probe_begin_batch(s);  // (the probe pool's clock: tables this batch uses are not evicted for one another)
  if (!opt_exhaustive && ((kUseBShare && n_bool_queries) || and_probe) && s->opt.use_dense && s->opt.dense && s->opt.probe_budget_x > 0) {
    bool built = false;
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const tq_query &q = queries[qi];
      // (2-term intersections of a batch large enough for the shared launch: the list that is probed — the
      // more frequent one — the same way; lists 2.. of longer intersections are reached through the term
      // table, which only knows a list's own tables)
      const bool and2 = and_probe && q.mode == TQ_MODE_AND && q.n_terms == 2;
      if (!((q.mode == TQ_MODE_BOOL && kUseBShare) || and2) || !q.terms || q.n_terms > TQD_AS_MAX_TERMS || q.k > 128u) continue;
      uint32_t skip = 0xFFFFFFFFu;  // (the leader of an intersection is decoded, never probed)
      if (and2 && q.terms[0] < s->terms.size() && q.terms[1] < s->terms.size())
        skip = s->terms[q.terms[1]].doc_freq < s->terms[q.terms[0]].doc_freq ? 1u : 0u;
      for (uint32_t i = 0; i < q.n_terms; ++i) {
        if (i == skip) continue;
        const uint32_t h = q.terms[i];
        if (h >= s->terms.size()) continue;  // (absent, or reported by plan_query)
        const TermHost &th = s->terms[h];
        if (th.dense_blob && th.tf8_blob) continue;
        if (and2 && th.rdir_blob && s->rdir_span_ok) continue;  // (probed through its range directory)

        // (a list too short for a directory — rdir_plan — is too short for a max_doc / 4-byte slot of the probe pool:
        // its queries keep the per-query kernel, whose leader has at most as many blocks)
        if (and2 && s->opt.rdir_budget_x > 0 && th.doc_freq < 256u) continue;
        const bool had = th.probe_dense_blob && th.probe_tf8_blob;
        bool ok = false;
        const int prc = build_probe_tables(s, h, &ok);  // (tables it already has: marked as used by this batch)
        if (prc != TQ_OK) return prc;
        built = built || (ok && !had);
      }
    }
    if (built) s->share_span_terms = ~(size_t)0;  // (the tables' address span is taken again below)
  }

  // nested boolean queries (tq_tree.hip) reach EVERY list through a bitmap, in both modes
  if (n_bool_queries && s->opt.use_dense && s->opt.dense && s->opt.probe_budget_x > 0) {
    bool built = false;
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const tq_query &q = queries[qi];
      if (q.mode != TQ_MODE_BOOL || !q.terms || q.n_terms > TQ_MAX_TERMS || !bool_query_is_tree(q)) continue;
      for (uint32_t i = 0; i < q.n_terms; ++i) {
        const uint32_t h = q.terms[i];
        if (h >= s->terms.size()) continue;
        const TermHost &th = s->terms[h];
        if (!(th.dense_blob && th.tf8_blob)) {
          const bool had = th.probe_dense_blob && th.probe_tf8_blob;
          bool ok = false;
          const int prc = build_probe_tables(s, h, &ok, true);
          if (prc != TQ_OK) return prc;
          built = built || (ok && !had);
        }
        // a term of a phrase inside the boolean query: its positions are reached from the bitmap's rank
        if (q.nested_occurs && q.nested_occurs[i] != 255u && (q.nested_occurs[i] & TQ_NESTED_PHRASE) &&
            !s->terms[h].posdir_blob && !s->terms[h].probe_posdir_blob) {
          bool ok = false;
          const int prc = build_probe_posdir(s, h, &ok);
          if (prc != TQ_OK) return prc;
          built = built || ok;
        }
      }
    }
    if (built) s->share_span_terms = ~(size_t)0;
  }

  // ---- plan
  const bool or_windows_opt = s->opt.or_windows < 0 ? opt_exhaustive != 0 : s->opt.or_windows != 0;
  // launch groups: AND queries whose non-leader lists all have a bitmap run a leaner kernel
  // instantiation (no seek / block-search code: fewer registers, less LDS, more waves per CU)
  // boolean queries (clauses with roles) run the candidate-driven union kernel's BOOL instantiation
  constexpr int kGroups = kNGroups, kAndGeneral = 3, kBool = 4, kShare = 5, kPhSweep = 6, kDense = 7, kAShare = 8, kBShare = 9, kTree = 10;
  // AND queries whose other lists all have a bitmap + byte-wide tfs, pruned, k <= 128, <= 8 lists, on a
  // segment with a doc matrix, whose leader (rarest list) leads at least kAShareMin such queries of the
  // batch: the shared-intersection launch (leader-major, tq_ashare.hip).  TQ_ASHARE=0: the per-query kernel
  static const bool kUseAShare = tune_u32("TQ_ASHARE", 1) != 0;
  // (round 4 asked for 4 queries per leader: the others ran on the per-query kernel NEXT to the shared launch, and two
  // kernel families sharing the chip cost more than lone leads in the shared launch do — 1 024 queries 1.39 -> 1.17 ms,
  // 2 048 1.63 -> 1.40, 4 096 1.95 -> 1.72, the mixed stream 7.6 -> 7.3 ms per batch with every qualifying query in it)
  static const uint32_t kAShareMin = std::max<uint32_t>(1u, tune_u32("TQ_AS_MIN_LEADS", 1));
  // phrases whose lists ALL have a bitmap, byte-wide tfs and a position directory, the rarest one
  // still about a posting per bitmap word: the bitmap-AND sweep (phrase_sweep_kernel)
  static const uint32_t kPhSweepRatio = tune_u32("TQ_PH_SWEEP_RATIO", 128);  // 0 = never (64 until round 6: with leaders down to max_doc / 128 the realistic phrase stream ran 10 % faster)
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
  update_table_span();  // (probe tables built above moved it)
  Group(&groups)[kGroups] = s->plan->groups;
  for (Group &g : groups) g.reset();
  groups[kBool].mode = TQ_MODE_OR;
  groups[kShare].mode = TQ_MODE_OR;
  groups[kPhSweep].mode = TQ_MODE_PHRASE;
  groups[kDense].mode = TQ_MODE_OR;
  groups[kAShare].mode = TQ_MODE_AND;
  groups[kBShare].mode = TQ_MODE_OR;
  groups[kTree].mode = TQ_MODE_OR;
  s->plan->xrow_term.clear();
  s->plan->xrow_of.clear();
  groups[0].mode = TQ_MODE_AND;
  groups[1].mode = TQ_MODE_OR;
  groups[2].mode = TQ_MODE_PHRASE;
  groups[kAndGeneral].mode = TQ_MODE_AND;
  uint64_t algo_bytes = 0;
  uint32_t n_thr_rows = 0;
  bool phrase_all_dense = true;
  // Which list would lead an AND query in the shared-intersection launch (0xFFFFFFFF: the query does not
  // qualify), and how many queries of the batch every list would lead; each distinct list's bytes once
  // (tq_batch_stats.unique_bytes: what the batch needs from the index when nothing is read twice).
  const bool ashare_on = kUseAShare && !opt_exhaustive && s->d_docmat && s->opt.use_dense && s->share_span_ok;
  bool ashare_and = ashare_on && !co.no_ashare;  // (intersections: also needs "ashare_min_batch" qualifying queries)
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
      if (!(th.dense_blob && th.tf8_blob) && !(q.n_terms == 2 && th.rdir_blob && s->rdir_span_ok) &&
          !(and_probe && q.n_terms == 2 && th.probe_dense_blob && th.probe_tf8_blob))
        return 0xFFFFFFFFu;
    }
    return q.terms[best_i];
  };
  uint64_t unique_bytes = 0;
  uint64_t and_lead_blocks = 0;  // leader blocks of the batch's intersections (sizes their tiles below)
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
      uint32_t lead_blocks = 0xFFFFFFFFu;
      for (uint32_t i = 0; i < q.n_terms; ++i) {
        const uint32_t h = q.terms[i];
        if (h >= s->terms.size()) continue;
        lead_blocks = std::min(lead_blocks, s->terms[h].n_blocks);
        if (ps.term_stamp[2 * h] != ps.batch_stamp) {
          ps.term_stamp[2 * h] = ps.batch_stamp;
          unique_bytes += s->terms[h].postings_len;
        }
        if (q.mode == TQ_MODE_PHRASE && ps.term_stamp[2 * h + 1] != ps.batch_stamp) {
          ps.term_stamp[2 * h + 1] = ps.batch_stamp;
          unique_bytes += s->terms[h].positions_len;
        }
      }
      if (q.mode == TQ_MODE_AND && q.n_terms >= 2 && lead_blocks != 0xFFFFFFFFu) and_lead_blocks += lead_blocks;
      if (ashare_on) {
        const uint32_t lh = ashare_leader(q, ps.q_cache[qi]);
        ps.q_leader[qi] = lh;
        if (lh != 0xFFFFFFFFu) ++ps.and_lead_count[lh];
      }
    }
    if (ashare_on) {  // a small batch keeps the per-query kernels ("ashare_min_batch")
      static const uint32_t kMinBatchEnv = tune_u32("TQ_AS_MIN_BATCH", 0xFFFFFFFFu);
      const uint32_t min_batch = kMinBatchEnv != 0xFFFFFFFFu ? kMinBatchEnv : (uint32_t)s->opt.ashare_min_batch;
      uint32_t n_el = 0;
      for (uint32_t qi = 0; qi < n_queries && n_el < min_batch; ++qi) {
        const uint32_t lh = ps.q_leader[qi];
        if (lh != 0xFFFFFFFFu && ps.and_lead_count[lh] >= kAShareMin) ++n_el;
      }
      ashare_and = n_el >= min_batch;
    }
  }
  // Tiles of the per-query intersection kernel: 64 leader blocks — unless the whole batch would then be a few hundred
  // wavefronts walking their tiles serially (one query over a 250 k-doc leader = 31 wavefronts x 64 blocks = 0.23 of
  // a 0.27 ms call): a batch of fewer than kSmallBatchBlocks leader blocks (one to four queries) is cut into about
  // kSmallBatchTiles tiles of at least kMinTileBlocks blocks, and the hundreds of partial lists that makes are merged in
  // two levels (tqk_launch_merge).  One query per call: 0.27 -> 0.14 ms.
  static const uint32_t kSmallBatchTiles = std::max<uint32_t>(1u, tune_u32("TQ_AND_SMALL_TILES", 2048));
  static const uint32_t kMinTileBlocks = std::min<uint32_t>(TQD_AND_TILE, std::max<uint32_t>(1u, tune_u32("TQ_AND_MIN_TILE", 4)));
  static const uint64_t kSmallBatchBlocks = tune_u32("TQ_AND_SMALL_BLOCKS", 8192);  // (tq_plan_chunks.cpp: the chunk floor follows)
  const uint32_t and_tile_cap = and_lead_blocks < kSmallBatchBlocks
                                    ? (uint32_t)std::min<uint64_t>(TQD_AND_TILE, std::max<uint64_t>(kMinTileBlocks, and_lead_blocks / kSmallBatchTiles))
                                    : TQD_AND_TILE;
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
    bool ph_sweep = false, ashare = false, bshare = false;
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
          dq.tile_blocks = std::min<uint32_t>(and_tile_cap, std::max<uint32_t>(1u, kAndTileNum / c_lb));
          tile_cost = dq.tile_blocks * c_lb;
          n_tiles = (lead_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
          bool nonneg = true;
          for (uint32_t i = 0; i < q.n_terms; ++i) nonneg = nonneg && dq.weight[i] >= 0.0f;
          if (ashare_and && nonneg && (all_dense || q.n_terms == 2)) {  // (2 lists: list 1 may have probe tables, q_leader knows)
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
          // A phrase matches a doc at most min(tf of its terms) times: bm25(sum of idfs, norm, min tf) bounds the doc's
          // score BEFORE its positions are read.  The reference filters finished scores only (weight.rs:47-60); with
          // the query's threshold shared across its wavefronts (the slots of the other pruned kernels) the phrase
          // kernels skip the position walk of every candidate that cannot reach the top-k — the same top-k, bit for bit
          // (candidates EQUAL to the bound are walked: ties resolve by doc id).
          static const bool kPhPrune = tune_u32("TQ_PH_PRUNE", 1) != 0;
          if (kPhPrune && !opt_exhaustive && q.weights[0] >= 0.0f && q.k <= TQD_THR_SLOTS) {
            dq.flags |= TQD_QF_PRUNE;
            dq.thr_index = n_thr_rows++;
          }
        }
      }
    }
    bool bool_done = false, share = false, dense_u = false, tree = false;
    if (q.mode == TQ_MODE_BOOL && bool_query_is_tree(q)) {
      // a clause that is itself a BooleanQuery of terms (intersection inside a union / under MustNot, nested
      // minimums: tq_query.nested_occurs): evaluated over the lists' bitmaps, tq_tree.hip
      if (!s->share_span_ok || !s->opt.use_dense)
        return fail(TQ_ERR_UNSUPPORTED,
                    "query %u: nested boolean queries need the lists' bitmaps (\"use_dense\" %d, one 32 GB table span: %s — tables from %llx, arena %llx + %zu MB "
                    "mapped, %zu allocations outside it)",
                    qi, (int)s->opt.use_dense, s->share_span_ok ? "ok" : "exceeded", (unsigned long long)s->share_table_lo, (unsigned long long)(uintptr_t)s->dense_arena,
                    s->dense_arena_mapped >> 20, s->dense_extra.size());
      TqdTreeQuery tq;
      const int rc = plan_tree_query(s, q, qi, tq, qbytes, s->share_table_lo);
      if (rc != TQ_OK) return rc;
      tq.cache_idx = cache_idx;
      groups[kTree].tree.push_back(tq);
      dq.n_terms = tq.n_terms;
      mode = TQ_MODE_OR;
      n_tiles = 0;
      bool_done = true;
      tree = true;
    } else if (q.mode == TQ_MODE_BOOL) {
      const int rc = plan_bool_query(s, q, qi, dq, qbytes, n_tiles, tile_cost, n_thr_rows, opt_exhaustive != 0);
      if (rc != TQ_OK) return rc;
      mode = TQ_MODE_OR;  // runs in the union launch group
      bool_done = true;
      // the shared launch (tq_ashare.hip, boolean leads): every list reached through its bitmap and tf bytes
      // (the only list of a lead set of one is only ever decoded)
      bshare = kUseBShare && !co.no_bshare && ashare_on && (dq.flags & TQD_QF_PRUNE) && dq.thr_index != 0xFFFFFFFFu && dq.n_terms >= 1 &&
               dq.n_terms <= TQD_AS_MAX_TERMS && q.k <= 128u && ps_plan.q_cache[qi] < 256u && dq.n_lead >= 1;
      for (uint32_t i = 0; bshare && i < dq.n_terms; ++i) {
        const TermHost &th = s->terms[dq.term[i]];
        if (!(th.dense_blob && th.tf8_blob) && !(th.probe_dense_blob && th.probe_tf8_blob) && !(kBsRdir && th.rdir_blob && s->rdir_span_ok) &&
            !(dq.n_lead == 1 && i == 0))
          bshare = false;
      }
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
        // (a batch with fewer unions than the launch needs never builds plain lists for it: those stay
        // charged to the segment's table budget for good — ADVICE r03)
        dense_u = kDenseRatio && n_union_queries >= kDenseMinQueries && opt_exhaustive && s->opt.use_dense && dq.n_terms >= 1 && dq.n_terms <= 8 &&
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
    Group &g = groups[tree ? kTree : bool_done ? (bshare ? kBShare : kBool) : (share ? kShare : (dense_u ? kDense : (ph_sweep ? kPhSweep : (ashare ? kAShare : ((mode == TQ_MODE_AND && !all_dense) ? kAndGeneral : mode)))))];
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
      for (uint32_t sb = 0; sb < q_slabs; ++sb) g.tree.insert(g.tree.end(), qs[sb].groups[gi].tree.begin(), qs[sb].groups[gi].tree.end());
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
    if (&g == &groups[kTree]) {  // one partial list per (query, tile of bitmap words); no chunk tables
      const uint32_t tiles = tqk_tree_tiles((s->max_doc + 31u) / 32u);
      g.kpl = kpl_for(g.max_k);
      for (TqdQuery &dq : g.queries) dq.n_parts = tiles;
      g.n_chunks = g.total_tiles = (uint32_t)g.queries.size() * tiles;
      continue;
    }
    const int crc = &g == &groups[kShare]    ? build_share_plan(s, g, *s->plan)
                    : &g == &groups[kAShare] ? build_ashare_plan(s, g, *s->plan, false)
                    : &g == &groups[kBShare] ? build_ashare_plan(s, g, *s->plan, true)
                    : &g == &groups[kDense]  ? build_dense_plan(s, g, *s->plan, (uint32_t)std::max(1, cus))
                                             : build_group_chunks(g, or_windows_opt && &g != &groups[kBool], *s->plan,
                                                                  &g == &groups[kBool]);
    // result lists of a shared-intersection launch over the budget even with the longest tasks: the same batch
    // through the per-query kernels (nothing has been enqueued yet)
    if (crc == TQ_ERR_UNSUPPORTED && (&g == &groups[kAShare] || &g == &groups[kBShare]) &&
        s->plan->ap[&g == &groups[kBShare] ? 1 : 0].over_budget) {
      CallOpts co2 = co;
      (&g == &groups[kBShare] ? co2.no_bshare : co2.no_ashare) = true;
      return search_batch_impl(s, queries, n_queries, out_stride, d_out_scores, d_out_docs, d_out_counts, hip_stream, co2);
    }
    if (crc != TQ_OK) return crc;
  }
  // partial lists of all groups share one buffer; its stride is per group (kpl*64 keys)
  size_t part_off_bytes[kGroups] = {};
  for (int gi = 0; gi < kGroups; ++gi) {
    Group &g = groups[gi];
    part_off_bytes[gi] = partial_bytes;
    if (gi == kShare || gi == kDense || gi == kAShare || gi == kBShare) {  // result lists: part_start / n_parts count 8-byte entries (build_share_plan)
      if (!g.queries.empty())  // (list_entries: the last query of a group may read an earlier query's list)
        partial_bytes += std::max<size_t>((size_t)g.list_entries, (size_t)g.queries.back().part_start + g.queries.back().n_parts) * sizeof(uint64_t);
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
    if (&g == &groups[kAShare] || &g == &groups[kBShare]) {
      const PlanScratch::ASharePlan &A = s->plan->ap[&g == &groups[kBShare] ? 1 : 0];
      stage = (stage + 63) & ~(size_t)63;
      g.o_leads = stage;
      stage += A.aleads.size() * sizeof(TqdALead);
      stage = (stage + 15) & ~(size_t)15;
      g.o_tasks = stage;
      stage += A.atasks.size() * sizeof(uint4);
      g.o_lists = stage;
      if (&g == &groups[kBShare]) stage += A.alists.size() * sizeof(uint2);
    }
    if (&g == &groups[kTree]) {  // (o_leads: the tree descriptors)
      stage = (stage + 63) & ~(size_t)63;
      g.o_leads = stage;
      stage += g.tree.size() * sizeof(TqdTreeQuery);
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
  stage = (stage + 15) & ~(size_t)15;
  const size_t o_terms = stage;
  stage += (terms_to - terms_from) * sizeof(TqdTerm);
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
  if (terms_to > terms_from) memcpy(hs + o_terms, s->h_dterms.data() + terms_from, (terms_to - terms_from) * sizeof(TqdTerm));
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
    if (&g == &groups[kAShare] || &g == &groups[kBShare]) {
      const PlanScratch::ASharePlan &A = s->plan->ap[&g == &groups[kBShare] ? 1 : 0];
      memcpy(hs + g.o_leads, A.aleads.data(), A.aleads.size() * sizeof(TqdALead));
      big_copy(hs + g.o_tasks, A.atasks.data(), A.atasks.size() * sizeof(uint4));
      if (&g == &groups[kBShare]) memcpy(hs + g.o_lists, A.alists.data(), A.alists.size() * sizeof(uint2));
    }
    if (&g == &groups[kTree]) {
      for (size_t i = 0; i < g.tree.size(); ++i) g.tree[i].part_start = g.queries[i].part_start;
      memcpy(hs + g.o_leads, g.tree.data(), g.tree.size() * sizeof(TqdTreeQuery));
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
  if (terms_to > terms_from) {  // the new term records: staging blob -> the table's tail, in stream order before the kernels
    HIP_TRY(hipMemcpyAsync(s->d_terms + terms_from, (const uint8_t *)dstage.p + o_terms, (terms_to - terms_from) * sizeof(TqdTerm),
                           hipMemcpyDeviceToDevice, st));
    s->d_terms_dirty = false;
    s->d_terms_dirty_from = ~(size_t)0;
    s->d_terms_synced = terms_to;
  }
  TqkZeroParams zp{};  // every buffer the batch zeroes, cleared by one launch below
  auto zero_later = [&](void *ptr, size_t bytes) -> int {
    if (!bytes) return TQ_OK;
    if (zp.n == (uint32_t)TQK_ZERO_MAX || (bytes & 3u) || ((uintptr_t)ptr & 3u) || bytes > 0xFFFFFFFFull * 4u) {
      HIP_TRY(hipMemsetAsync(ptr, 0, bytes, st));  // (never taken today: at most six aligned buffers)
      return TQ_OK;
    }
    zp.ptr[zp.n] = (uint32_t *)ptr;
    zp.words[zp.n++] = (uint32_t)(bytes >> 2);
    return TQ_OK;
  };
  rc = zero_later(s->d_match_counter, sizeof(unsigned long long));
  if (rc == TQ_OK) rc = zero_later(s->d_qmatches.p, (size_t)n_queries * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  s->last_batch_queries = n_queries;
  if (n_thr_rows) {
    const size_t thr_bytes = (size_t)n_thr_rows * TQD_THR_SLOTS * sizeof(uint32_t);
    rc = s->d_thr.ensure(thr_bytes);
    if (rc != TQ_OK) return rc;
    // TQ_KEEP_THR=1 (experiments only): the slots keep the previous batch's final values, i.e. the
    // same batch run again starts from its final thresholds (what perfect threshold knowledge buys)
    static const bool kKeepThr = tune_u32("TQ_KEEP_THR", 0) != 0;
    if (!kKeepThr || !s->thr_seeded) {
      rc = zero_later(s->d_thr.p, thr_bytes);
      if (rc != TQ_OK) return rc;
    }
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
    rc = zero_later(s->d_share_words.p, words * sizeof(uint32_t));
    if (rc != TQ_OK) return rc;
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
    rc = zero_later(s->d_share_words.p, words * sizeof(uint32_t));
    if (rc != TQ_OK) return rc;
  }

  // the shared-intersection launches: [0] intersections, [1] boolean queries
  const int a_group[2] = {kAShare, kBShare};
  uint32_t ashare_grid[2] = {0, 0};
  const size_t n_ashare_of[2] = {groups[kAShare].queries.size(), groups[kBShare].queries.size()};
  const size_t n_ashare = n_ashare_of[0];
  for (int ai = 0; ai < 2; ++ai) {
    if (!n_ashare_of[ai]) continue;  // thr_val | list_count per query, then the task counters; staging lists of the persistent grid
    static const uint32_t kAGridMul = tune_u32("TQ_AS_GRID_MUL", 0);
    const uint32_t per_cu = kAGridMul ? kAGridMul : (ai ? tqk_bshare_waves_per_cu() : tqk_ashare_waves_per_cu());
    ashare_grid[ai] = (uint32_t)std::min<uint64_t>(groups[a_group[ai]].n_chunks, (uint64_t)std::max(1, cus) * per_cu);
    const size_t words = 2 * n_ashare_of[ai] + 16;  // (+ 8 task counters per launch)
    DevBuf &wb = ai ? s->d_bshare_words : s->d_ashare_words;
    rc = wb.ensure(words * sizeof(uint32_t));
    if (rc == TQ_OK)
      rc = (ai ? sc.bshare_stage : sc.ashare_stage)
               .ensure((size_t)ashare_grid[ai] * TQD_AS_GROUP * tqk_share_capl(groups[a_group[ai]].kpl) * sizeof(uint64_t));
    if (rc != TQ_OK) return rc;
    // (TQ_KEEP_THR=1, experiments only: the threshold words keep the previous batch's final values — what a
    // launch costs whose thresholds are right from its first task)
    static const bool kKeepThrWords = tune_u32("TQ_KEEP_THR", 0) != 0;
    const size_t keep = (kKeepThrWords && s->thr_seeded) ? n_ashare_of[ai] : 0;
    rc = zero_later((uint32_t *)wb.p + keep, (words - keep) * sizeof(uint32_t));
    if (rc != TQ_OK) return rc;
  }
  {
    const hipError_t ze = tqk_launch_zero(zp, st);
    if (ze != hipSuccess) return fail(TQ_ERR_HIP, "zero launch: %s", hipGetErrorString(ze));
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
  const int launch_order[kGroups] = {kAndGeneral, kBool, kBShare, kShare, kDense, kTree, 1, 2, kPhSweep, 0, kAShare};  // long serial chains first
  // the group that keeps the caller's stream: the batch's intersections
  // (a batch without either: the phrase sweep, else the first group that has queries — two groups that both went to the
  // side stream ran one after the other: the few sparse-leader phrases of a phrase batch added their 0.26 ms)
  int main_group = n_ashare ? kAShare : 0;
  if (groups[main_group].queries.empty()) {
    if (!groups[kPhSweep].queries.empty()) {
      main_group = kPhSweep;
    } else {
      for (int gi = 0; gi < kGroups; ++gi)
        if (!groups[gi].queries.empty()) {
          main_group = gi;
          break;
        }
    }
  }
  uint32_t kernel_mask = 0;
  for (int oi = 0; oi < kGroups; ++oi) {
    const int gi = launch_order[oi];
    Group &g = groups[gi];
    if (g.queries.empty()) continue;
    // the big dense-AND group keeps the caller's stream, the others go to the side stream
    hipStream_t gst = (fork && gi != main_group) ? s->side_stream : st;
    if (gi == kAShare || gi == kBShare) {
      const int ai = gi == kBShare ? 1 : 0;
      const size_t n_a = n_ashare_of[ai];
      TqkAShareParams ap{};
      ap.seg = s->dseg;
      ap.terms = s->d_terms;
      ap.queries = (const TqdQuery *)(ds + g.o_queries);
      ap.caches = (const float *)(ds + o_caches);
      ap.leads = (const TqdALead *)(ds + g.o_leads);
      ap.tasks = (const uint4 *)(ds + g.o_tasks);
      ap.qlists = (const uint2 *)(ds + g.o_lists);
      ap.sinks = (const TqkSinks *)(ds + g.o_sinks);
      ap.thr_slots = (uint32_t *)s->d_thr.p;
      ap.thr_val = (uint32_t *)(ai ? s->d_bshare_words : s->d_ashare_words).p;
      ap.list_count = ap.thr_val + n_a;
      ap.table_base = (const uint8_t *)s->plan->share_table_base;
      ap.stage = (uint64_t *)(ai ? sc.bshare_stage : sc.ashare_stage).p;
      ap.lists = (uint64_t *)((uint8_t *)sc.partials.p + part_off_bytes[gi]);
      ap.n_queries = (uint32_t)n_a;
      ap.boolean = (uint32_t)ai;
      ap.rdir_lists = ai && s->plan->ap[ai].any_rdir ? 1u : 0u;
      static const uint32_t kDebugA = tune_u32("TQ_DEBUG", 0);
      ap.debug = s->opt.debug >= 0 ? (uint32_t)s->opt.debug : kDebugA;  // (option "debug": work counters of one batch, bench.py)
      ap.bound_slack = co.bound_slack;
      // TQ_AS_BOUND (A/B of the bound on the non-leader list of an intersection; every setting is exact):
      // bit 0 the block pre-filter takes list 1's range maxima over the leader block's doc span, bit 1 the
      // gather cut of a block comes from that pre-filter, bit 2 the per-(doc, lead) test reads the doc's own
      // range (measured slower: two more dependent byte gathers per (block, lead) pair cost more than the 7 % of
      // scoring-stage candidates they remove); 0 = round 4: list 1 is bounded by its weight.  Default 3.
      static const uint32_t kBoundMode = tune_u32("TQ_AS_BOUND", 3);
      ap.bound_mode = kBoundMode;
      tiles_total += g.total_tiles;
      chunks_total += g.n_chunks;
      kernel_mask |= ai ? TQ_KERNEL_BSHARE : TQ_KERNEL_ASHARE;
      // two launches: the warm-up tasks, then the rest (stream order = the barrier between them)
      const uint32_t bounds[3] = {0u, s->plan->ap[ai].a_warm_tasks, g.n_chunks};
      for (int ph = 0; ph < 2; ++ph) {
        ap.task_begin = bounds[ph];
        ap.n_tasks = bounds[ph + 1];
        if (ap.n_tasks <= ap.task_begin) continue;
        // (TQ_AS_QUEUES=8: one task queue per XCD, each a contiguous eighth of the doc-slice order — measured
        // SLOWER than one queue, 1.67 against 1.50 ms for the headline batch, 2.06 against 1.72 at 4096 terms)
        static const uint32_t kQueues = std::min<uint32_t>(8u, std::max<uint32_t>(1u, tune_u32("TQ_AS_QUEUES", 1)));
        ap.n_queues = ph ? kQueues : 1u;
        ap.task_counter = ap.thr_val + 2 * n_a + 8 * ph;
        ap.grid = std::min<uint32_t>(ashare_grid[ai], ap.n_tasks - ap.task_begin);
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
      sp.debug = s->opt.debug >= 0 ? (uint32_t)s->opt.debug : kDebugS;
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
    if (gi == kTree) {
      kernel_mask |= TQ_KERNEL_TREE;
      TqkTreeParams tp{};
      tp.seg = s->dseg;
      tp.terms = s->d_terms;
      tp.queries = (const TqdTreeQuery *)(ds + g.o_leads);
      tp.caches = (const float *)(ds + o_caches);
      tp.sinks = (const TqkSinks *)(ds + g.o_sinks);
      tp.table_base = (const uint8_t *)s->share_table_lo;
      tp.n_queries = (uint32_t)g.queries.size();
      tp.n_words = (s->max_doc + 31u) / 32u;
      for (const TqdTreeQuery &tq : g.tree) tp.any_phrase |= tq.has_phrase;
      tiles_total += g.total_tiles;
      chunks_total += g.n_chunks;
      const hipError_t e = tqk_launch_tree(tp, g.kpl, gst);
      if (e != hipSuccess) return fail(TQ_ERR_HIP, "nested boolean launch: %s", hipGetErrorString(e));
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
    if (!s->opt.use_dense) p.seg.docmat = nullptr, p.seg.doccls = nullptr;
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
    p.debug = s->opt.debug >= 0 ? (uint32_t)s->opt.debug : kDebug;
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
  if (s->opt.record_query_kernels) {  // (diagnosis: which family ran which query, tq_last_batch_query_kernels)
    s->last_query_kernel.assign(n_queries, 0u);
    for (int gi = 0; gi < kGroups; ++gi) {
      const Group &g = groups[gi];
      if (g.queries.empty()) continue;
      const uint32_t bit = gi == kAShare ? TQ_KERNEL_ASHARE : gi == kBShare ? TQ_KERNEL_BSHARE : gi == kShare ? TQ_KERNEL_USHARE
                           : gi == kDense ? TQ_KERNEL_XUNION : gi == kTree ? TQ_KERNEL_TREE : gi == 0 ? TQ_KERNEL_AND_DENSE : gi == kAndGeneral ? TQ_KERNEL_AND
                           : gi == kBool ? TQ_KERNEL_BOOL : gi == kPhSweep ? TQ_KERNEL_PHRASE_SWEEP : gi == 2 ? TQ_KERNEL_PHRASE
                           : ((or_windows_opt && gi != kBool) ? TQ_KERNEL_OR_WINDOWS : TQ_KERNEL_UNION);
      for (uint32_t qi : g.out_index) s->last_query_kernel[qi] = bit;
    }
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
    if (gi != kAShare && gi != kBShare && gi != kShare && gi != kDense) {
      // a query with hundreds of partial lists (a small batch cut into short tiles): its lists are reduced by up
      // to 32 wavefronts first — one wavefront walking 2 000 lists was most of a one-query call
      uint32_t max_parts = 0;
      for (const TqdQuery &dq : g.queries) max_parts = std::max(max_parts, dq.n_parts);
      if (max_parts >= TQK_MERGE_PRE_MIN && m.n_queries <= 4096u)
        m.pre_slices = std::min<uint32_t>(32u, std::max<uint32_t>(2u, (uint32_t)std::sqrt((double)max_parts)));
    }
    hipError_t e = gi == kAShare  ? tqk_launch_merge_lists(m, (const uint32_t *)s->d_ashare_words.p + n_ashare, g.kpl, st)
                   : gi == kBShare ? tqk_launch_merge_lists(m, (const uint32_t *)s->d_bshare_words.p + n_ashare_of[1], g.kpl, st)
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
    for (int gi = 0; gi < kGroups; ++gi)
      if (!groups[gi].queries.empty())
        fprintf(stderr, "[tq] group %d: %zu queries, %u tasks\n", gi, groups[gi].queries.size(), groups[gi].n_chunks);
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
  // (atomic loads: tq_submit resolves a call's options without the segment lock)
  co.exhaustive = __atomic_load_n(&s->opt.exhaustive, __ATOMIC_RELAXED) != 0;
  uint32_t ppm = (uint32_t)__atomic_load_n(&s->opt.bound_slack_ppm, __ATOMIC_RELAXED);
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
                      uint32_t *out_counts, const CallOpts &co) {
  if (!s || !out_scores || !out_docs || !out_counts)
    return fail(TQ_ERR_INVALID, "tq_search_batch: null argument");
  if (n_queries == 0) return TQ_OK;
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = (size_t)n_queries * out_stride;
  // The merge kernels write the rows STRAIGHT into pinned host memory (hipHostMalloc memory is mapped into the
  // device's address space): scores | docs | counts in one buffer, no device-to-host copy operations — three copies into
  // the caller's pageable arrays were three staged, blocking operations (and an asynchronous copy into pinned memory
  // now and then held the calling thread for 7 ms, tantivy_amd/distributed.py).  TQ_HOST_OUT_COPY=1: the copies.
  static const bool kCopyOut = tune_u32("TQ_HOST_OUT_COPY", 0) != 0;
  if (!kCopyOut) {
    const size_t o_docs = (n * sizeof(float) + 255) & ~(size_t)255;
    const size_t o_counts = (o_docs + n * sizeof(uint32_t) + 255) & ~(size_t)255;
    int rc = s->h_out.ensure(o_counts + (size_t)n_queries * sizeof(uint32_t));
    if (rc != TQ_OK) return rc;
    uint8_t *h = (uint8_t *)s->h_out.p;
    rc = search_batch_impl(s, queries, n_queries, out_stride, (float *)h, (uint32_t *)(h + o_docs),
                           (uint32_t *)(h + o_counts), nullptr, co);
    if (rc != TQ_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    memcpy(out_scores, h, n * sizeof(float));
    memcpy(out_docs, h + o_docs, n * sizeof(uint32_t));
    memcpy(out_counts, h + o_counts, (size_t)n_queries * sizeof(uint32_t));
    return TQ_OK;
  }
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
int search_batch_host_begin(tq_segment *s, const tq_query *queries, uint32_t n_queries, uint32_t out_stride,
                            const CallOpts &co, HostBatchSlot &slot) {
  if (!s || !n_queries) return fail(TQ_ERR_INVALID, "tq_submit: empty batch");
  HIP_TRY(hipSetDevice(s->device));
  const size_t n = (size_t)n_queries * out_stride;
  slot.o_docs = (n * sizeof(float) + 255) & ~(size_t)255;
  slot.o_counts = (slot.o_docs + n * sizeof(uint32_t) + 255) & ~(size_t)255;
  slot.n = n_queries;
  slot.stride = out_stride;
  int rc = slot.out.ensure(slot.o_counts + (size_t)n_queries * sizeof(uint32_t));
  if (rc != TQ_OK) return rc;
  if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
  if (!slot.done_blocking) HIP_TRY(hipEventCreateWithFlags(&slot.done_blocking, hipEventDisableTiming | hipEventBlockingSync));
  uint8_t *h = (uint8_t *)slot.out.p;
  rc = search_batch_impl(s, queries, n_queries, out_stride, (float *)h, (uint32_t *)(h + slot.o_docs),
                         (uint32_t *)(h + slot.o_counts), nullptr, co);
  if (rc != TQ_OK) return rc;
  // A batch of many callers is waited for asleep (an interrupt, some tens of microseconds later than a spinning wait
  // would notice): with hundreds of request threads the host's cores are what runs out first — the GPU box grants 16,
  // and a process above its quota is stopped for the rest of the 100 ms period, all its threads (p99 of 1 024 threads:
  // 92 ms) — and a core spinning for 0.7 ms per batch is one the callers do not have.  A small batch is a few callers
  // waiting for exactly this: the spinning wait.
  static const uint32_t kBlockMin = tune_u32("TQ_SUBMIT_BLOCK_MIN", 128);
  slot.blocking = n_queries >= kBlockMin;
  HIP_TRY(hipEventRecord(slot.blocking ? slot.done_blocking : slot.done, s->stream));
  return TQ_OK;
}
int search_batch_host_end(tq_segment *s, HostBatchSlot &slot) {
  (void)s;
  HIP_TRY(hipEventSynchronize(slot.blocking ? slot.done_blocking : slot.done));
  return TQ_OK;
}
}  // namespace tqi

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
  try {  // (the planner allocates: nothing may unwind across the C boundary)
    TQ_SEGMENT_LOCK(s);
    CallOpts co;
    const int rc = resolve_opts(s, opts, co);
    if (rc != TQ_OK) return rc;
    return search_batch_impl(s, queries, n_queries, out_stride, d_out_scores, d_out_docs,
                             d_out_counts, hip_stream, co);
  } catch (const std::exception &e) {
    return fail(TQ_ERR_HIP, "tq_search_batch_device: %s", e.what());
  } catch (...) {
    return fail(TQ_ERR_HIP, "tq_search_batch_device: unknown exception");
  }
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
  try {
    TQ_SEGMENT_LOCK(s);
    CallOpts co;
    const int rc = resolve_opts(s, opts, co);
    if (rc != TQ_OK) return rc;
    return search_batch_host(s, queries, n_queries, out_stride, out_scores, out_docs, out_counts, co);
  } catch (const std::exception &e) {
    return fail(TQ_ERR_HIP, "tq_search_batch: %s", e.what());
  } catch (...) {
    return fail(TQ_ERR_HIP, "tq_search_batch: unknown exception");
  }
}

}  // extern "C"
