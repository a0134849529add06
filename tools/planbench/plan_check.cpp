// Host-only invariants of the chunk planner (build_group_chunks of tq_api.cpp), driven by
// tests/test_planner_cpu.py: synthetic launch groups (candidate unions with per-leader tile runs,
// AND-style single runs), planned with 1 and with N threads.  Checked for every plan:
//   * the chunks tile [0, total_tiles) exactly once, in order;
//   * a chunk's query is the query that holds its first tile;
//   * chunk_recs is a permutation of the chunks (every chunk launched exactly once);
//   * chunk_first / n_parts of a query = the chunks that hold its tiles.
#include "../../tantivy_amd/csrc/tq_internal.hpp"

#include <random>

static int fail_msg(const char *m, long a = 0, long b = 0) {
  fprintf(stderr, "plan_check: %s (%ld, %ld)\n", m, a, b);
  return 1;
}

static int check(Group &g, bool or_win) {
  const uint32_t n = g.n_chunks;
  if (g.chunk_recs.size() != n) return fail_msg("chunk_recs size", (long)g.chunk_recs.size(), n);
  std::vector<uint4> by_chunk(n, make_uint4(0, 0, 0, 0xFFFFFFFFu));
  for (uint32_t b = 0; b < n; ++b) {
    const uint4 r = g.chunk_recs[b];
    if (r.w >= n) return fail_msg("chunk id out of range", r.w, n);
    if (by_chunk[r.w].w != 0xFFFFFFFFu) return fail_msg("chunk launched twice", r.w);
    by_chunk[r.w] = r;
    if (or_win && r.w != b) return fail_msg("window kernel: launch order must be chunk order", b, r.w);
  }
  uint32_t expect = 0;
  for (uint32_t c = 0; c < n; ++c) {
    const uint4 r = by_chunk[c];
    if (r.x != expect) return fail_msg("chunks do not tile the range", c, r.x);
    if (r.y <= r.x) return fail_msg("empty chunk", c);
    expect = r.y;
    const TqdQuery &q = g.queries[r.z];
    if (r.x < q.tile_start || r.x >= q.tile_start + q.n_tiles) return fail_msg("first tile not in its query", c, r.z);
  }
  if (expect != g.total_tiles) return fail_msg("last chunk does not end at total_tiles", expect, g.total_tiles);
  const uint32_t per_chunk = or_win ? TQD_WAVES_PER_WG : 1u;
  for (size_t qi = 0; qi < g.queries.size(); ++qi) {
    const TqdQuery &q = g.queries[qi];
    if (!q.n_tiles) continue;
    uint32_t first = 0xFFFFFFFFu, last = 0;
    for (uint32_t c = 0; c < n; ++c)
      if (by_chunk[c].y > q.tile_start && by_chunk[c].x < q.tile_start + q.n_tiles) {
        if (first == 0xFFFFFFFFu) first = c;
        last = c;
      }
    // n_parts counts from the query's first chunk to the last chunk built while planning it
    if (q.chunk_first != first) return fail_msg("chunk_first", (long)qi, q.chunk_first);
    if (q.n_parts < (last - first + 1) * per_chunk) return fail_msg("n_parts too small", (long)qi, q.n_parts);
  }
  return 0;
}

// The shared-union planner (build_share_plan): every (query, list) pair is exactly one lead; the
// tasks of a (position, term) run cover the term's blocks exactly once per lead group, in position
// order; every lead record describes its query (columns, signature bits, weights); the result-list
// regions are disjoint and hold k entries per (task, lead) pair.
static int check_share(int seed) {
  std::mt19937 rng(seed + 100);
  auto uni = [&](uint32_t lo, uint32_t hi) { return std::uniform_int_distribution<uint32_t>(lo, hi)(rng); };
  tq_segment seg;
  const uint32_t n_terms = 300;
  static uint8_t arena[1 << 20];
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    th.n_blocks = uni(1, 3000);
    th.doc_freq = th.n_blocks * 128u;
    TqdTerm dt{};
    dt.has_freq = 1u;
    if (t < 80) {  // dense: bitmap + tf8 at fake (distinct, 8-aligned) addresses; the first 40 have columns
      th.dense_blob = arena + 4096u * t + 8u;
      th.tf8_blob = arena + 4096u * t + 2048u;
      if (t < TQD_MAT_SLOTS) dt.has_freq |= (t + 1u) << 8;
    }
    if (t >= TQD_MAT_SLOTS && t % 3) dt.has_freq |= ((t * 7u) % TQD_SIG_BITS + 1u) << 16;  // signature bits for some
    seg.terms.push_back(th);
    seg.h_dterms.push_back(dt);
  }
  seg.max_doc = 10000000u;
  PlanScratch ps;
  Group &g = ps.groups[5];
  g.reset();
  g.mode = TQ_MODE_OR;
  const uint32_t nq = 700;
  for (uint32_t q = 0; q < nq; ++q) {
    TqdQuery dq{};
    dq.n_terms = uni(1, TQD_US_MAX_TERMS);
    dq.k = uni(1, 128);
    dq.flags = TQD_QF_PRUNE;
    dq.cache_idx = uni(0, 2);
    dq.thr_index = 4u * q;
    uint32_t used[TQD_US_MAX_TERMS];
    for (uint32_t i = 0; i < dq.n_terms; ++i) {
      uint32_t t;
      bool dup;
      do {
        t = uni(0, n_terms - 1);
        dup = false;
        for (uint32_t j = 0; j < i; ++j) dup |= used[j] == t;
      } while (dup);
      used[i] = t;
      dq.term[i] = t;
      dq.weight[i] = 30.0f / (float)(i + 1);  // descending
    }
    g.queries.push_back(dq);
    g.tile_cost.push_back(1);
    g.out_index.push_back(q);
    g.max_k = std::max(g.max_k, dq.k);
  }
  seg.share_table_lo = (uint64_t)arena;  // (what search_batch_impl computes over the segment's terms)
  if (build_share_plan(&seg, g, ps) != TQ_OK) return fail_msg("build_share_plan failed");
  size_t n_leads = 0;
  for (const TqdQuery &q : g.queries) n_leads += q.n_terms;
  if (ps.leads.size() != n_leads) return fail_msg("lead count", (long)ps.leads.size(), (long)n_leads);
  std::vector<uint8_t> seen(nq * TQD_US_MAX_TERMS, 0);
  for (const TqdLead &ld : ps.leads) {
    const uint32_t li = ld.info & 15u, nt = (ld.info >> 8) & 15u;
    if (ld.query >= nq || li >= g.queries[ld.query].n_terms) return fail_msg("lead out of range", ld.query, li);
    const TqdQuery &q = g.queries[ld.query];
    if (nt != q.n_terms || seen[ld.query * TQD_US_MAX_TERMS + li]++) return fail_msg("lead twice / n_terms", ld.query, li);
    if (ld.w != q.weight[li]) return fail_msg("lead weight", ld.query, li);
    float suffix = 0;
    for (uint32_t m = q.n_terms; m-- > li;) suffix += q.weight[m];
    if (ld.suffix != suffix) return fail_msg("lead suffix", ld.query, li);
    uint32_t c = 0;
    float sparse = 0;
    for (uint32_t m = 0; m < q.n_terms; ++m) {
      const uint32_t hf = seg.h_dterms[q.term[m]].has_freq;
      const uint32_t col = (hf >> 8) & 0xFFu ? 8u + ((hf >> 8) & 0xFFu) - 1u : 0u;
      const uint32_t sig1 = col ? 0u : (hf >> 16) & 0xFFu;
      if (((ld.info >> (16 + m)) & 1u) != (col ? 0u : 1u)) return fail_msg("nocol bit", ld.query, m);
      if (((ld.info >> (24 + m)) & 1u) != ((col || sig1) ? 0u : 1u)) return fail_msg("nopc bit", ld.query, m);
      if (ld.sig[m] != sig1) return fail_msg("sig byte", ld.query, m);
      if (m < li && col && !((ld.before_mask >> col) & 1ull)) return fail_msg("before_mask", ld.query, m);
      if (m > li) {
        const uint32_t want = col ? col : (sig1 ? TQD_SIG_SHIFT + sig1 - 1u : 0u);
        if (want) {
          const uint32_t got = ((c < 4 ? ld.cols_lo >> (8 * c) : ld.cols_hi >> (8 * (c - 4)))) & 0xFFu;
          if (got != want || ld.aw[c] != q.weight[m]) return fail_msg("column entry", ld.query, m);
          ++c;
        } else {
          sparse += q.weight[m];
        }
        const TermHost &th = seg.terms[q.term[m]];
        const uint64_t d = th.dense_blob ? (uint64_t)th.dense_blob - ps.share_table_base : 0;
        if ((uint64_t)ld.dense_off[m - li - 1] * 8u != d) return fail_msg("dense_off", ld.query, m);
      }
    }
    if (((ld.info >> 4) & 15u) != c || ld.sparse_after != sparse) return fail_msg("ncols / sparse_after", ld.query, li);
  }
  // tasks: per (lead group) the runs of blocks tile the term's list; positions ascend
  std::vector<uint32_t> pairs(nq, 0);
  std::unordered_map<uint64_t, uint32_t> next_block;  // (first lead << 8 | n_leads) -> next expected block
  uint32_t last_pos = 0;
  for (size_t ti = 0; ti < ps.tasks.size(); ++ti) {
    const uint4 t = ps.tasks[ti];
    const uint32_t nb = t.z & 0xFFFFu, nl = (t.z >> 16) & 0xFFu, cache = t.z >> 24;
    if (!nb || !nl || nl > TQD_US_GROUP || t.w + nl > ps.leads.size()) return fail_msg("task shape", (long)ti);
    uint32_t pos = 0;
    for (uint32_t l = 0; l < nl; ++l) {
      const TqdLead &ld = ps.leads[t.w + l];
      const TqdQuery &q = g.queries[ld.query];
      if (q.term[ld.info & 15u] != t.x || q.cache_idx != cache) return fail_msg("task lead of another term", (long)ti, l);
      pos = ld.info & 15u;
      ++pairs[ld.query];
    }
    if (pos < last_pos) return fail_msg("positions not ascending", (long)ti);
    last_pos = pos;
    if ((uint32_t)ti < ps.share_phase_first[pos] || (uint32_t)ti >= ps.share_phase_first[pos + 1])
      return fail_msg("phase bounds", (long)ti, pos);
    uint32_t &nx = next_block[((uint64_t)t.w << 8) | nl];
    if (t.y != nx) return fail_msg("runs do not tile the list", (long)ti, t.y);
    nx += nb;
    if (nx > seg.terms[t.x].n_blocks) return fail_msg("run past the list", (long)ti);
  }
  for (auto &kv : next_block) {
    const TqdLead &ld = ps.leads[kv.first >> 8];
    if (kv.second != seg.terms[g.queries[ld.query].term[ld.info & 15u]].n_blocks) return fail_msg("list not covered");
  }
  uint64_t at = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    if (g.queries[q].part_start != at) return fail_msg("result regions overlap", q);
    if (g.queries[q].n_parts != pairs[q] * g.queries[q].k) return fail_msg("result region size", q);
    at += g.queries[q].n_parts;
  }
  printf("share: %u queries, %zu leads, %zu tasks ok\n", nq, ps.leads.size(), ps.tasks.size());
  return 0;
}

// The shared-intersection planner (build_ashare_plan): every query is exactly one lead; the leads of
// one (leader, cache) are contiguous and sorted by mask; every lead record describes its query; the
// tasks of a lead group tile the leader's blocks exactly once; tasks are launched in doc-slice
// order; the result-list regions are disjoint and hold k entries per (task, lead) pair.
static int check_ashare(int seed) {
  std::mt19937 rng(seed + 500);
  auto uni = [&](uint32_t lo, uint32_t hi) { return std::uniform_int_distribution<uint32_t>(lo, hi)(rng); };
  tq_segment seg;
  const uint32_t n_terms = 120;
  static uint8_t arena[1 << 20];
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    th.n_blocks = uni(1, 30000) >> (t / 8);
    if (!th.n_blocks) th.n_blocks = 1;
    th.doc_freq = th.n_blocks * 128u - uni(0, 127);
    TqdTerm dt{};
    dt.has_freq = 1u;
    th.dense_blob = arena + 4096u * t + 8u;
    th.tf8_blob = arena + 4096u * t + 2048u;
    if (t < TQD_MAT_SLOTS) dt.has_freq |= (t + 1u) << 8;
    else if (t % 5) dt.has_freq |= ((t * 7u) % TQD_SIG_BITS + 1u) << 16;  // (some lists with neither)
    seg.terms.push_back(th);
    seg.h_dterms.push_back(dt);
  }
  seg.max_doc = 10000000u;
  PlanScratch ps;
  Group &g = ps.groups[8];
  g.reset();
  g.mode = TQ_MODE_AND;
  const uint32_t nq = uni(1, 4) == 1 ? uni(1, 40) : uni(500, 3000);
  for (uint32_t q = 0; q < nq; ++q) {
    TqdQuery dq{};
    dq.n_terms = uni(0, 3) ? 2u : uni(2, TQD_AS_MAX_TERMS);
    dq.k = uni(1, 128);
    dq.flags = TQD_QF_PRUNE;
    dq.cache_idx = uni(0, 9) ? 0u : uni(1, 2);
    dq.thr_index = 4u * q;
    if (q && uni(0, 2) == 0) {  // a third of the queries repeat an earlier one (twins), some with another k
      dq = g.queries[uni(0, q - 1)];
      dq.thr_index = 4u * q;
      if (uni(0, 1)) dq.k = uni(1, 128);
      g.queries.push_back(dq);
      g.tile_cost.push_back(1);
      g.out_index.push_back(q);
      g.max_k = std::max(g.max_k, dq.k);
      continue;
    }
    uint32_t used[TQD_AS_MAX_TERMS];
    for (uint32_t i = 0; i < dq.n_terms; ++i) {
      uint32_t t;
      bool dup;
      do {
        t = uni(0, 1) ? uni(0, 15) : uni(0, n_terms - 1);
        dup = false;
        for (uint32_t j = 0; j < i; ++j) dup |= used[j] == t;
      } while (dup);
      used[i] = t;
    }
    std::sort(used, used + dq.n_terms, [&](uint32_t a, uint32_t b) {  // doc freq ascending: term[0] leads
      return seg.terms[a].doc_freq != seg.terms[b].doc_freq ? seg.terms[a].doc_freq < seg.terms[b].doc_freq : a < b;
    });
    for (uint32_t i = 0; i < dq.n_terms; ++i) {
      dq.term[i] = used[i];
      dq.weight[i] = 1.0f + 0.37f * (float)((used[i] * 13u + q) % 11u);
    }
    g.queries.push_back(dq);
    g.tile_cost.push_back(1);
    g.out_index.push_back(q);
    g.max_k = std::max(g.max_k, dq.k);
  }
  seg.share_table_lo = (uint64_t)arena;
  const uint64_t list_budget = (uint64_t)std::max<uint32_t>(1u, tune_u32("TQ_AS_LIST_MB", 1024)) << 20;
  if (build_ashare_plan(&seg, g, ps) != TQ_OK) {
    // the only failure a well-formed group may see: result lists over the budget with every run at the longest
    // task the kernel takes (511 blocks: a lead's docs scored per task are counted in 16 bits) — the caller
    // then plans the batch without the shared launch
    if (!ps.ap[0].over_budget) return fail_msg("build_ashare_plan failed");
    uint64_t floor_entries = 0;  // what the lists need at 511-block tasks, no warm-up
    for (uint32_t q = 0; q < nq; ++q)
      floor_entries += (uint64_t)((seg.terms[g.queries[q].term[0]].n_blocks + 510u) / 511u + 1u) * g.queries[q].k;  // (+ warm-up tasks, at least)
    if (floor_entries * 8u <= list_budget) return fail_msg("over_budget reported, but the longest tasks fit");
    printf("ashare: %u queries over the %llu MB list budget at the longest tasks (%llu entries): refused ok\n", nq,
           (unsigned long long)(list_budget >> 20), (unsigned long long)floor_entries);
    return 0;
  }
  // identical queries (lists, weights, k, cache) are evaluated once: the first of them owns the leads and the
  // result list, the others name it in chunk_first (what merge_lists_kernel reads)
  auto identical = [&](const TqdQuery &a, const TqdQuery &b) {
    return a.n_terms == b.n_terms && a.k == b.k && a.cache_idx == b.cache_idx && !memcmp(a.term, b.term, a.n_terms * 4) &&
           !memcmp(a.weight, b.weight, a.n_terms * 4);
  };
  uint32_t n_heads = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    const uint32_t o = g.queries[q].chunk_first;
    if (o > q || o >= nq) return fail_msg("list owner out of range", q, o);
    if (o == q) {
      ++n_heads;
      const bool dedupe = !getenv("TQ_AS_DEDUPE") || atoi(getenv("TQ_AS_DEDUPE")) != 0;  // (0: every query keeps its own lead)
      for (uint32_t e = 0; dedupe && e < q; ++e)
        if (g.queries[e].chunk_first == e && identical(g.queries[e], g.queries[q])) return fail_msg("an identical query before this one was not made its owner", q, e);
      continue;
    }
    if (g.queries[o].chunk_first != o) return fail_msg("the owner of a list is not a head", q, o);
    if (!identical(g.queries[o], g.queries[q])) return fail_msg("list owner is another query", q, o);
    if (g.queries[q].part_start != g.queries[o].part_start || g.queries[q].n_parts != g.queries[o].n_parts)
      return fail_msg("a repeated query does not read its owner's list", q, o);
  }
  if (ps.ap[0].aleads.size() != n_heads) return fail_msg("lead count", (long)ps.ap[0].aleads.size(), n_heads);
  std::vector<uint8_t> seen(nq, 0);
  uint64_t prev_key = 0, prev_mask = 0;
  for (size_t i = 0; i < ps.ap[0].aleads.size(); ++i) {
    const TqdALead &ld = ps.ap[0].aleads[i];
    if (ld.query >= nq || seen[ld.query]++) return fail_msg("lead twice / out of range", (long)i, ld.query);
    if (g.queries[ld.query].chunk_first != ld.query) return fail_msg("a lead for a query that reads another's list", (long)i, ld.query);
    const TqdQuery &q = g.queries[ld.query];
    if ((ld.info & 31u) != q.n_terms || ld.w != q.weight[0] || ld.k != q.k) return fail_msg("lead header", (long)i);
    float rest = 0;
    uint64_t mask = 0;
    for (uint32_t m = 1; m < q.n_terms; ++m) {
      rest += q.weight[m];
      const uint32_t hf = seg.h_dterms[q.term[m]].has_freq;
      const uint32_t col = (hf >> 8) & 0xFFu ? 8u + ((hf >> 8) & 0xFFu) - 1u : 0u;
      const uint32_t sig1 = col ? 0u : (hf >> 16) & 0xFFu;
      if (col) mask |= 1ull << col;
      else if (sig1) mask |= 1ull << (TQD_SIG_SHIFT + sig1 - 1u);
    }
    if (ld.rest != rest) return fail_msg("lead rest", (long)i);
    if (q.n_terms == 2 && ld.rest != q.weight[1]) return fail_msg("2-term rest is list 1's weight", (long)i);
    if ((((uint64_t)ld.mask_hi << 32) | ld.mask_lo) != mask) return fail_msg("lead mask", (long)i);
    const bool col1 = ((seg.h_dterms[q.term[1]].has_freq >> 8) & 0xFFu) != 0u;
    if (((ld.info >> 8) & 1u) != (col1 ? 1u : 0u)) return fail_msg("column flag", (long)i);
    const TermHost &t1 = seg.terms[q.term[1]];
    if ((uint64_t)ld.dense_off * 8u != (uint64_t)t1.dense_blob - ps.share_table_base ||
        (uint64_t)ld.tf8_off * 8u != (uint64_t)t1.tf8_blob - ps.share_table_base)
      return fail_msg("table offsets", (long)i);
    const uint64_t key = ((uint64_t)q.term[0] << 8) | q.cache_idx;
    // (TQ_AS_DEDUPE=1: by the planner's bin of the mask; =0: by the mask itself — either way equal masks are neighbours)
    const bool by_bin = !getenv("TQ_AS_DEDUPE") || atoi(getenv("TQ_AS_DEDUPE")) != 0;
    const uint64_t mk = by_bin ? alead_mask_bin(mask) : mask, prev_mk = by_bin ? alead_mask_bin(prev_mask) : prev_mask;
    if (i && (key < prev_key || (key == prev_key && mk < prev_mk))) return fail_msg("lead order", (long)i);
    prev_key = key;
    prev_mask = mask;
  }
  std::vector<uint32_t> pairs(nq, 0);
  std::unordered_map<uint64_t, uint32_t> next_block;  // (first lead << 8 | n_leads) -> next expected block
  std::unordered_map<uint64_t, uint32_t> covered;     // lead index -> tasks that name it (via its group)
  uint32_t last_slice = 0;
  for (size_t ti = 0; ti < ps.ap[0].atasks.size(); ++ti) {
    const uint4 t = ps.ap[0].atasks[ti];
    const uint32_t nb = t.z & 0xFFFFu, nl = (t.z >> 16) & 0xFFu, cache = t.z >> 24;
    if (!nb || !nl || nl > TQD_AS_GROUP || t.w + nl > ps.ap[0].aleads.size()) return fail_msg("task shape", (long)ti);
    for (uint32_t l = 0; l < nl; ++l) {
      const TqdALead &ld = ps.ap[0].aleads[t.w + l];
      const TqdQuery &q = g.queries[ld.query];
      if (q.term[0] != t.x || q.cache_idx != cache) return fail_msg("task lead of another leader", (long)ti, l);
      ++pairs[ld.query];
      // twin bit: set iff the lead is the same query (lists, weights) as the one before it in the group
      bool same = false;
      if (l) {
        const TqdQuery &pq = g.queries[ps.ap[0].aleads[t.w + l - 1].query];
        same = pq.n_terms == q.n_terms && pq.k == q.k && !memcmp(pq.term, q.term, q.n_terms * 4) &&
               !memcmp(pq.weight, q.weight, q.n_terms * 4);
        if (same && ps.ap[0].aleads[t.w + l - 1].thr_row != ld.thr_row) return fail_msg("twins share one row of threshold slots", (long)ti, l);
      }
      if ((((ld.info >> 9) & 1u) != 0u) != same) return fail_msg("twin bit", (long)ti, l);
    }
    const uint32_t n_blocks = seg.terms[t.x].n_blocks;
    // the warm-up launch (tasks [0, a_warm_tasks): the first blocks of every leader), then doc-slice order
    const bool warm = ti < ps.ap[0].a_warm_tasks;
    // (the planner's slice of a run: (first block << 12) / n_blocks by a multiplication with 2^44 / n_blocks)
    const uint32_t slice = warm ? 0u : 1u + std::min<uint32_t>(4095u, (uint32_t)((t.y * (((uint64_t)1 << 44) / n_blocks)) >> 32));
    if (slice < last_slice) return fail_msg("tasks not in doc-slice order", (long)ti);
    last_slice = slice;
    // (tasks of one group appear in block order: the sort by slice is stable and slices follow blocks)
    uint32_t &nx = next_block[((uint64_t)t.w << 8) | nl];
    if (t.y != nx) return fail_msg("runs do not tile the list", (long)ti, t.y);
    nx += nb;
    if (nx > n_blocks) return fail_msg("run past the list", (long)ti);
  }
  size_t leads_in_groups = 0;
  for (auto &kv : next_block) {
    const TqdALead &ld = ps.ap[0].aleads[kv.first >> 8];
    if (kv.second != seg.terms[g.queries[ld.query].term[0]].n_blocks) return fail_msg("list not covered");
    leads_in_groups += kv.first & 0xFFu;
  }
  if (leads_in_groups != n_heads) return fail_msg("lead groups do not partition the leads", (long)leads_in_groups, n_heads);
  uint64_t at = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    if (g.queries[q].chunk_first != q) {
      if (pairs[q]) return fail_msg("a repeated query has tasks of its own", q);
      continue;
    }
    if (!pairs[q]) return fail_msg("query without a task", q);
    if (g.queries[q].part_start != at) return fail_msg("result regions overlap", q);
    if (g.queries[q].n_parts != pairs[q] * g.queries[q].k) return fail_msg("result region size", q);
    at += g.queries[q].n_parts;
  }
  if (g.list_entries != at) return fail_msg("list_entries", (long)g.list_entries);
  if (g.n_chunks != ps.ap[0].atasks.size()) return fail_msg("n_chunks");
  if (at * 8u > list_budget) return fail_msg("result lists over the budget", (long)at);
  printf("ashare: %u queries, %zu tasks, %llu list entries ok\n", nq, ps.ap[0].atasks.size(), (unsigned long long)at);
  return 0;
}

// The boolean leads of the shared launch (build_ashare_plan(.., boolean = true)): one lead per (query,
// list of its lead set); the tasks of a lead group tile the leader's blocks exactly once; result
// regions hold k entries per (task, lead) pair of the query.  And what the device's filter stage does
// with a lead's masks is SAFE: for random membership vectors (doc-matrix words with exact column bits and
// signature bits that may be set for non-members), a doc that the query matches and that belongs to this
// lead is never dropped by the masks, and its bound (leader weight + the weights of the lists its word
// does not rule out) is never below the score the doc can reach; a list the lead marks "never probed"
// never holds such a doc.
static int check_bshare(int seed) {
  std::mt19937 rng(seed + 900);
  auto uni = [&](uint32_t lo, uint32_t hi) { return std::uniform_int_distribution<uint32_t>(lo, hi)(rng); };
  tq_segment seg;
  const uint32_t n_terms = 90;
  static uint8_t arena[1 << 20];
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    th.n_blocks = std::max<uint32_t>(1u, uni(1, 20000) >> (t / 8));
    th.doc_freq = th.n_blocks * 128u - uni(0, 127);
    TqdTerm dt{};
    dt.has_freq = 1u;
    th.dense_blob = arena + 4096u * t + 8u;
    th.tf8_blob = arena + 4096u * t + 2048u;
    if (t < TQD_MAT_SLOTS) dt.has_freq |= (t + 1u) << 8;
    else if (t % 5) dt.has_freq |= ((t * 7u) % TQD_SIG_BITS + 1u) << 16;  // (some lists with neither)
    seg.terms.push_back(th);
    seg.h_dterms.push_back(dt);
  }
  seg.max_doc = 10000000u;
  seg.share_table_lo = (uint64_t)arena;
  PlanScratch ps;
  Group &g = ps.groups[9];
  g.reset();
  g.mode = TQ_MODE_OR;
  const uint32_t nq = uni(1, 3) == 1 ? uni(1, 30) : uni(300, 1500);
  for (uint32_t q = 0; q < nq; ++q) {
    TqdQuery dq{};
    if (q && uni(0, 3) == 0) {  // repeats (twins)
      dq = g.queries[uni(0, q - 1)];
    } else {
      // plan_bool_query's layout: [optional leading Should | lead Must clause | Must clauses | MustNot | Should]
      const bool has_must = uni(0, 3) != 0;
      const uint32_t n_opt = has_must ? uni(0, 2) : 0u;
      const uint32_t n_leadc = has_must ? uni(1, 2) : uni(1, 4);
      uint32_t n = 0;
      auto put = [&](uint32_t role) {
        uint32_t t;
        bool dup;
        do {
          t = uni(0, 2) ? uni(0, 47) : uni(0, n_terms - 1);
          dup = false;
          for (uint32_t j = 0; j < n; ++j) dup |= dq.term[j] == t;
        } while (dup);
        dq.term[n] = t;
        dq.weight[n] = role == TQD_ROLE_MUST_NOT ? 0.0f : 0.5f + 0.25f * (float)uni(0, 12);
        dq.roles |= role << (2u * n);
        ++n;
      };
      for (uint32_t i = 0; i < n_opt; ++i) put(TQD_ROLE_SHOULD);
      dq.n_opt_lead = n_opt;
      for (uint32_t i = 0; i < n_leadc; ++i) put(has_must ? TQD_ROLE_MUST : TQD_ROLE_SHOULD);
      dq.n_lead = n;
      if (has_must) {
        const uint32_t n_clauses = uni(0, 2);
        for (uint32_t c = 0; c < n_clauses && n + 2 < TQD_AS_MAX_TERMS; ++c) {
          const uint32_t w = uni(1, 2);
          for (uint32_t i = 0; i < w; ++i) put(TQD_ROLE_MUST);
          dq.clause_end |= 1u << (n - 1u);
        }
      }
      for (uint32_t i = uni(0, 2); i > 0 && n < TQD_AS_MAX_TERMS; --i) put(TQD_ROLE_MUST_NOT);
      if (has_must && !n_opt)
        for (uint32_t i = uni(0, 2); i > 0 && n < TQD_AS_MAX_TERMS; --i) put(TQD_ROLE_SHOULD);
      dq.n_terms = n;
      dq.cache_idx = uni(0, 9) ? 0u : 1u;
      dq.flags = TQD_QF_PRUNE;
    }
    dq.k = uni(1, 128);
    dq.thr_index = 4u * q;
    g.queries.push_back(dq);
    g.tile_cost.push_back(1);
    g.out_index.push_back(q);
    g.max_k = std::max(g.max_k, dq.k);
  }
  if (build_ashare_plan(&seg, g, ps, true) != TQ_OK) {
    if (!ps.ap[1].over_budget) return fail_msg("build_ashare_plan (boolean) failed");
    printf("bshare: %u queries over the result-list budget at the longest tasks: refused ok\n", nq);  // (the caller plans without the shared launch)
    return 0;
  }
  const PlanScratch::ASharePlan &A = ps.ap[1];
  size_t want_leads = 0;  // (a query identical to an earlier one reads that query's list: chunk_first names it)
  for (uint32_t q = 0; q < nq; ++q) {
    const TqdQuery &dq = g.queries[q];
    if (dq.chunk_first == q) {
      want_leads += dq.n_lead;
      continue;
    }
    const TqdQuery &o = g.queries[dq.chunk_first];
    if (dq.chunk_first > q || o.chunk_first != dq.chunk_first || o.n_terms != dq.n_terms || o.k != dq.k || o.roles != dq.roles ||
        o.clause_end != dq.clause_end || o.n_lead != dq.n_lead || o.n_opt_lead != dq.n_opt_lead || o.min_should != dq.min_should ||
        o.cache_idx != dq.cache_idx || memcmp(o.term, dq.term, dq.n_terms * 4) || memcmp(o.weight, dq.weight, dq.n_terms * 4) ||
        o.part_start != dq.part_start || o.n_parts != dq.n_parts)
      return fail_msg("a repeated boolean query and the owner of its list differ", q, dq.chunk_first);
  }
  if (A.aleads.size() != want_leads) return fail_msg("lead count", (long)A.aleads.size(), (long)want_leads);
  auto bit_of = [&](uint32_t handle) -> uint32_t {
    const uint32_t hf = seg.h_dterms[handle].has_freq;
    const uint32_t slot1 = (hf >> 8) & 0xFFu, sig1 = (hf >> 16) & 0xFFu;
    return slot1 ? 8u + slot1 - 1u : (sig1 ? TQD_SIG_SHIFT + sig1 - 1u : 0u);
  };
  std::vector<uint32_t> seen(nq, 0);
  uint64_t prev_key = 0;
  for (size_t i = 0; i < A.aleads.size(); ++i) {
    const TqdALead &ld = A.aleads[i];
    if (ld.query >= nq) return fail_msg("lead out of range", (long)i);
    const TqdQuery &q = g.queries[ld.query];
    const uint32_t li = (ld.info >> 16) & 15u;
    if (li >= q.n_lead || (seen[ld.query] >> li) & 1u) return fail_msg("lead twice / not a lead-set list", (long)i, li);
    seen[ld.query] |= 1u << li;
    if ((ld.info & 31u) != q.n_terms || ld.w != q.weight[li] || ld.k != q.k) return fail_msg("lead header", (long)i);
    const uint64_t key = ((uint64_t)q.term[li] << 8) | q.cache_idx;
    if (i && key < prev_key) return fail_msg("lead order", (long)i);
    prev_key = key;
    for (uint32_t m = 0; m < q.n_terms; ++m) {
      const uint2 bl = A.alists[(size_t)ld.query * TQD_AS_MAX_TERMS + m];
      if ((uint64_t)bl.x * 8u != (uint64_t)seg.terms[q.term[m]].dense_blob - ps.share_table_base ||
          (uint64_t)bl.y * 8u != (uint64_t)seg.terms[q.term[m]].tf8_blob - ps.share_table_base)
        return fail_msg("list table", (long)i, m);
    }
    const uint64_t excl = ((uint64_t)ld.excl_hi << 32) | ld.excl_lo, any1 = ((uint64_t)ld.any1_hi << 32) | ld.any1_lo,
                   any2 = ((uint64_t)ld.any2_hi << 32) | ld.any2_lo, bytes = ((uint64_t)ld.tf8_off << 32) | ld.dense_off;
    float rest_base;
    memcpy(&rest_base, &ld.mask_lo, sizeof(float));
    const uint32_t maybe = (ld.info >> 20) & 0xFFu;
    // random docs of this lead's list
    for (int trial = 0; trial < 24; ++trial) {
      bool in[TQD_AS_MAX_TERMS];
      uint64_t word = 0;
      for (uint32_t m = 0; m < q.n_terms; ++m) {
        in[m] = m == li || uni(0, 2) == 0;
        const uint32_t b = bit_of(q.term[m]);
        if (b && (in[m] || (b >= TQD_SIG_SHIFT && uni(0, 3) == 0))) word |= 1ull << b;  // (signature bits: false positives)
      }
      word |= (uint64_t)uni(0, 0xFFFFu) << TQD_SIG_SHIFT & (uni(0, 1) ? ~0ull : 0ull);  // (other lists' signature bits)
      for (uint32_t b = 8; b < TQD_SIG_SHIFT; ++b) {  // other lists' columns, at random
        bool ours = false;
        for (uint32_t m = 0; m < q.n_terms; ++m) ours |= bit_of(q.term[m]) == b;
        if (!ours && uni(0, 3) == 0) word |= 1ull << b;
      }
      // does the query match the doc, and does the doc belong to this lead?
      bool match = true, lead_clause = q.n_opt_lead == q.n_lead;  // (no Must: the lead set is the union itself)
      float reach = 0.0f;
      bool cfound = false;
      for (uint32_t m = 0; m < q.n_terms; ++m) {
        const uint32_t role = (q.roles >> (2u * m)) & 3u;
        if (role == TQD_ROLE_MUST_NOT) {
          if (in[m]) match = false;
          continue;
        }
        if (m < li && in[m]) match = false;  // an earlier list of the lead set holds it: that lead's doc
        if (m >= q.n_opt_lead && m < q.n_lead && in[m]) lead_clause = true;
        if (m >= q.n_lead && role == TQD_ROLE_MUST) {
          cfound |= in[m];
          if ((q.clause_end >> m) & 1u) {
            if (!cfound) match = false;
            cfound = false;
          }
        }
        if (m > li && in[m]) reach += q.weight[m];
      }
      if (q.n_opt_lead < q.n_lead && !lead_clause) match = false;
      if (!match) continue;
      if (word & excl) return fail_msg("exclusion mask drops a match", (long)i, trial);
      if (any1 && !(word & any1)) return fail_msg("clause mask 1 drops a match", (long)i, trial);
      if (any2 && !(word & any2)) return fail_msg("clause mask 2 drops a match", (long)i, trial);
      float bound = rest_base;
      for (uint32_t m = 0; m < q.n_terms; ++m) {
        const uint32_t bp = (uint32_t)(bytes >> (8u * m)) & 0xFFu;
        if (bp && ((word >> bp) & 1ull)) bound += q.weight[m];
        const bool probed = ((maybe >> m) & 1u) && (!bp || ((word >> bp) & 1ull));
        if (m != li && in[m] && !probed) return fail_msg("a list that holds the doc is not probed", (long)i, m);
      }
      if (bound * 1.00001f < reach) return fail_msg("doc bound below its reach", (long)i, trial);
      if (ld.rest * 1.00001f < reach) return fail_msg("lead rest below a reach", (long)i, trial);
    }
  }
  for (uint32_t q = 0; q < nq; ++q)
    if (seen[q] != (g.queries[q].chunk_first == q ? (1u << g.queries[q].n_lead) - 1u : 0u)) return fail_msg("a lead-set list without a lead", q);
  // tasks: groups of leads x runs of blocks, every (lead, block) once
  std::vector<uint32_t> pairs(nq, 0);
  std::unordered_map<uint64_t, uint32_t> next_block;
  for (size_t ti = 0; ti < A.atasks.size(); ++ti) {
    const uint4 t = A.atasks[ti];
    const uint32_t nb = t.z & 0xFFFFu, nl = (t.z >> 16) & 0xFFu;
    if (!nb || !nl || nl > TQD_AS_GROUP || t.w + nl > A.aleads.size()) return fail_msg("task shape", (long)ti);
    for (uint32_t l = 0; l < nl; ++l) {
      const TqdALead &ld = A.aleads[t.w + l];
      const TqdQuery &q = g.queries[ld.query];
      if (q.term[(ld.info >> 16) & 15u] != t.x || q.cache_idx != (t.z >> 24)) return fail_msg("task lead mismatch", (long)ti, l);
      ++pairs[ld.query];
    }
    uint32_t &nx = next_block[((uint64_t)t.w << 8) | nl];
    if (t.y != nx) return fail_msg("runs do not tile the list", (long)ti, t.y);
    nx += nb;
    if (nx > seg.terms[t.x].n_blocks) return fail_msg("run past the list", (long)ti);
    if ((ti < A.a_warm_tasks) != (t.y < seg.terms[t.x].n_blocks * 2ull / 1000ull)) return fail_msg("warm-up split", (long)ti);
  }
  for (auto &kv : next_block) {
    const TqdALead &ld = A.aleads[kv.first >> 8];
    if (kv.second != seg.terms[g.queries[ld.query].term[(ld.info >> 16) & 15u]].n_blocks) return fail_msg("list not covered");
  }
  uint64_t at = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    if (g.queries[q].chunk_first != q) {
      if (pairs[q]) return fail_msg("a repeated query has tasks of its own", q);
      continue;
    }
    if (!pairs[q]) return fail_msg("query without a task", q);
    if (g.queries[q].part_start != at) return fail_msg("result regions overlap", q);
    if (g.queries[q].n_parts != pairs[q] * g.queries[q].k) return fail_msg("result region size", q);
    at += g.queries[q].n_parts;
  }
  printf("bshare: %u queries, %zu leads, %zu tasks, %llu list entries ok\n", nq, A.aleads.size(), A.atasks.size(), (unsigned long long)at);
  return 0;
}

// The doc-major union plan (build_dense_plan): rows = the distinct (list, weight) pairs, lists with a
// bitmap first; every query's row bytes lead back to its lists at its weights, padded with the
// all-zero row; tasks cover every tile once; result lists are disjoint.
static int check_dense(int seed) {
  std::mt19937 rng(seed + 300);
  auto uni = [&](uint32_t lo, uint32_t hi) { return std::uniform_int_distribution<uint32_t>(lo, hi)(rng); };
  tq_segment seg;
  const uint32_t n_terms = 200;
  static uint8_t arena[1 << 20];
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    th.doc_freq = uni(10, 4000000);
    th.n_blocks = (th.doc_freq + 127) / 128;
    if (t % 3 == 0) {
      th.dense_blob = arena + 4096u * t + 8u;
      th.tf8_blob = arena + 4096u * t + 2048u;
    } else {
      th.flat_blob = arena + 4096u * t + 16u;
    }
    seg.terms.push_back(th);
    seg.h_dterms.push_back(TqdTerm{});
  }
  seg.max_doc = uni(1, 3) == 1 ? uni(1, 5000) : uni(100000, 12000000);
  PlanScratch ps;
  Group &g = ps.groups[7];
  g.reset();
  g.mode = TQ_MODE_OR;
  const uint32_t nq = uni(1, 900);
  for (uint32_t q = 0; q < nq; ++q) {
    TqdQuery dq{};
    dq.n_terms = uni(1, 8);
    dq.k = uni(1, 128);
    dq.thr_index = 4u * q;
    for (uint32_t i = 0; i < dq.n_terms; ++i) {
      dq.term[i] = uni(0, 60);  // (few lists: the 255 rows are never exceeded, as the caller guarantees)
      dq.weight[i] = (dq.term[i] % 7 == 0 && uni(0, 1)) ? 2.5f : 1.0f + 0.01f * (float)dq.term[i];  // some lists at two weights
      const uint64_t key = xrow_key(dq.term[i], dq.weight[i]);
      if (ps.xrow_of.emplace(key, (uint32_t)ps.xrow_term.size()).second) ps.xrow_term.push_back(key);
    }
    g.queries.push_back(dq);
    g.tile_cost.push_back(1);
    g.out_index.push_back(q);
    g.max_k = std::max(g.max_k, dq.k);
  }
  const uint32_t cus = uni(1, 304);
  if (build_dense_plan(&seg, g, ps, cus) != TQ_OK) return fail_msg("build_dense_plan failed");
  const uint32_t n_rows = (uint32_t)ps.xrows.size();
  if (n_rows != ps.xrow_term.size() || n_rows >= TQK_XU_MAX_ROWS) return fail_msg("row count", n_rows);
  std::vector<uint64_t> seen;
  for (uint32_t r = 0; r < n_rows; ++r) {
    const TqkDenseRow &row = ps.xrows[r];
    const TermHost &th = seg.terms[row.handle];
    const bool bitmap = th.dense_blob && th.tf8_blob;
    if (bitmap != (r < ps.x_bitmap_rows)) return fail_msg("bitmap rows first", r, ps.x_bitmap_rows);
    if (bitmap ? ((const void *)row.dense != th.dense_blob || (const void *)row.tf8 != th.tf8_blob || row.flat_docs)
               : ((const void *)row.flat_docs != th.flat_blob || row.dense || !row.tf8))
      return fail_msg("row tables", r);
    if (row.doc_freq != th.doc_freq) return fail_msg("row doc_freq", r);
    seen.push_back(xrow_key(row.handle, row.w));
  }
  std::sort(seen.begin(), seen.end());
  if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return fail_msg("a (list, weight) pair twice");
  uint32_t max_terms = 1;
  uint64_t list_end = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    const TqdQuery &dq = g.queries[q];
    const TqkDenseQuery &xq = ps.xqueries[q];
    max_terms = std::max(max_terms, dq.n_terms);
    if ((xq.nt_k & 0xFFu) != dq.n_terms || (xq.nt_k >> 8) != dq.k || xq.thr_row != dq.thr_index) return fail_msg("query header", q);
    for (uint32_t i = 0; i < 8u; ++i) {
      const uint32_t row = ((i < 4 ? xq.rows_lo >> (8u * i) : xq.rows_hi >> (8u * (i - 4u)))) & 0xFFu;
      if (i >= dq.n_terms) {
        if (row != n_rows) return fail_msg("padding row", q, row);
        continue;
      }
      if (row >= n_rows || ps.xrows[row].handle != dq.term[i] || ps.xrows[row].w != dq.weight[i]) return fail_msg("query row", q, i);
    }
    if (dq.part_start != list_end || dq.n_parts != ps.x_list_stride) return fail_msg("result list", q);
    list_end += dq.n_parts;
  }
  if (ps.x_max_terms != max_terms) return fail_msg("max_terms", ps.x_max_terms, max_terms);
  const uint32_t n_tiles = (seg.max_doc + TQK_XU_TILE - 1) / TQK_XU_TILE;
  if (g.total_tiles != n_tiles || ps.x_tiles_per_task < 1 || ps.x_tiles_per_task > 32) return fail_msg("tiles", g.total_tiles, n_tiles);
  if ((uint64_t)g.n_chunks * ps.x_tiles_per_task < n_tiles || (uint64_t)(g.n_chunks - 1) * ps.x_tiles_per_task >= n_tiles)
    return fail_msg("tasks cover the tiles once", g.n_chunks, ps.x_tiles_per_task);
  if (ps.xgrid < 1 || ps.xgrid > cus || ps.xgrid > g.n_chunks) return fail_msg("grid", ps.xgrid, cus);
  if (ps.x_list_stride != ps.xgrid * g.max_k) return fail_msg("list stride", ps.x_list_stride);
  printf("dense: %u queries, %u rows (%u with a bitmap), %u tasks ok\n", nq, n_rows, ps.x_bitmap_rows, g.n_chunks);
  return 0;
}

// The planner's thread pool (parallel_slabs / PlanPool): every slab of every job runs exactly once,
// whatever the number of slabs, also when several host threads plan at the same time (the pool
// serves one of them, the others run their slabs themselves) and when jobs follow each other faster
// than the helpers go back to sleep.
static int check_pool(int seed) {
  std::atomic<long> bad{0};
  auto caller = [&](uint32_t who) {
    std::mt19937 rng(seed * 17 + who);
    for (int job = 0; job < 3000; ++job) {
      const uint32_t n = 1u + rng() % 23u;
      std::vector<uint32_t> hits(n, 0u);
      uint64_t sum = 0;
      std::atomic<uint64_t> acc{0};
      parallel_slabs(n, [&](uint32_t sb) {
        ++hits[sb];  // (a slab is touched by one thread only)
        acc.fetch_add((uint64_t)(sb + 1) * (who + 1), std::memory_order_relaxed);
      });
      for (uint32_t i = 0; i < n; ++i) {
        if (hits[i] != 1u) ++bad;
        sum += (uint64_t)(i + 1) * (who + 1);
      }
      if (acc.load() != sum) ++bad;
      if (job % 500 == 0) std::this_thread::sleep_for(std::chrono::microseconds(700));  // let the helpers fall asleep
    }
  };
  std::vector<std::thread> th;
  for (uint32_t w = 0; w < 4; ++w) th.emplace_back(caller, w);
  for (std::thread &t : th) t.join();
  if (bad.load()) return fail_msg("pool: slabs run other than exactly once", bad.load());
  printf("pool: 4 callers x 3000 jobs ok\n");
  return 0;
}

// Count over bitmap words (tq_count.cpp::count_expression + the expression count_bitmap_kernel evaluates,
// restated here word for word): for random AND / OR / boolean queries — nested unions, absent terms,
// minimum_number_should_match, lists with and without a bitmap of their own — over random doc sets, the
// expression's doc set equals BooleanWeight's: all Must clauses, no MustNot clause, at least
// max(msm, 1 if there is no Must clause) Should clauses.  Queries count_expression hands to the scan are
// only the ones that are no bitwise expression of this shape (m of n Should clauses with 2 <= m < n).
static int check_count(int seed) {
  std::mt19937 rng(seed + 1300);
  auto uni = [&](uint32_t lo, uint32_t hi) { return std::uniform_int_distribution<uint32_t>(lo, hi)(rng); };
  tq_segment seg;
  const uint32_t n_terms = 24, n_words = 8;  // 256 docs
  seg.max_doc = 32 * n_words;
  std::vector<std::vector<uint2>> wide(n_terms);      // lists with a bitmap of their own
  std::vector<std::vector<uint32_t>> bits(n_terms);   // every list's doc bits
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    bits[t].resize(n_words);
    uint32_t df = 0;
    for (uint32_t w = 0; w < n_words; ++w) {
      bits[t][w] = (uint32_t)rng() & (uint32_t)rng() & (t % 3 ? 0xFFFFFFFFu : (uint32_t)rng());
      df += (uint32_t)__builtin_popcount(bits[t][w]);
    }
    th.doc_freq = df ? df : 1;
    if (t % 2 == 0) {
      wide[t].resize(n_words);
      for (uint32_t w = 0; w < n_words; ++w) wide[t][w] = make_uint2(bits[t][w], 0u);
      th.dense_blob = wide[t].data();
      th.tf8_blob = wide[t].data();
    }
    seg.terms.push_back(th);
    seg.h_dterms.push_back(TqdTerm{});
  }
  uint32_t n_expr = 0, n_scan = 0, n_known = 0;
  for (int trial = 0; trial < 4000; ++trial) {
    tq_query q{};
    uint32_t terms[TQ_MAX_TERMS];
    uint8_t occurs[TQ_MAX_TERMS], clause_of[TQ_MAX_TERMS];
    float weights[TQ_MAX_TERMS];
    q.n_terms = uni(1, 7);
    q.mode = (uint8_t)(uni(0, 3) == 0 ? TQ_MODE_AND : (uni(0, 2) == 0 ? TQ_MODE_OR : TQ_MODE_BOOL));
    const bool nested = q.mode == TQ_MODE_BOOL && uni(0, 1);
    uint8_t occ_of_clause[TQ_MAX_TERMS];
    for (uint32_t c = 0; c < TQ_MAX_TERMS; ++c) occ_of_clause[c] = (uint8_t)uni(0, 2);
    for (uint32_t i = 0; i < q.n_terms; ++i) {
      terms[i] = uni(0, 9) == 0 ? TQ_TERM_ABSENT : uni(0, n_terms - 1);
      clause_of[i] = (uint8_t)(nested ? uni(0, 3) : i);
      occurs[i] = occ_of_clause[clause_of[i]];  // (one occur per clause)
      weights[i] = 1.0f;
    }
    q.terms = terms;
    q.weights = weights;
    q.occurs = q.mode == TQ_MODE_BOOL ? occurs : nullptr;
    q.clause_of = nested ? clause_of : nullptr;
    q.min_should_match = q.mode == TQ_MODE_BOOL ? (uni(0, 2) ? 0u : uni(1, 3)) : 0u;
    q.k = 1;
    // BooleanWeight's doc set, clause by clause
    struct Cl { uint32_t id, occur; std::vector<uint32_t> ts; };
    std::vector<Cl> cls;
    for (uint32_t i = 0; i < q.n_terms; ++i) {
      const uint32_t id = q.mode == TQ_MODE_BOOL ? clause_of[i] : i;
      const uint32_t oc = q.mode == TQ_MODE_AND ? TQ_MUST : (q.mode == TQ_MODE_OR ? TQ_SHOULD : occurs[i]);
      size_t c = 0;
      while (c < cls.size() && cls[c].id != id) ++c;
      if (c == cls.size()) cls.push_back(Cl{id, oc, {}});
      if (terms[i] != TQ_TERM_ABSENT) cls[c].ts.push_back(terms[i]);
    }
    uint32_t n_must = 0, n_should = 0;
    for (const Cl &c : cls) {
      if (c.occur == TQ_MUST) ++n_must;
      else if (c.occur == TQ_SHOULD && !c.ts.empty()) ++n_should;
    }
    const uint32_t need_should = std::max<uint32_t>(q.min_should_match, n_must ? 0u : 1u);
    uint32_t want = 0;
    for (uint32_t d = 0; d < seg.max_doc; ++d) {
      bool all_must = true, any_not = false;
      uint32_t sc = 0;
      for (const Cl &c : cls) {
        bool in = false;
        for (uint32_t t : c.ts) in = in || ((bits[t][d >> 5] >> (d & 31u)) & 1u);
        if (c.occur == TQ_MUST) all_must = all_must && in;
        else if (c.occur == TQ_MUST_NOT) any_not = any_not || in;
        else if (in) ++sc;
      }
      if (all_must && !any_not && sc >= need_should) ++want;
    }
    TqkCountQuery cq;
    bool known = false;
    uint64_t driver = 0;
    std::unordered_map<uint32_t, uint32_t> slots;
    const bool expr = count_expression(&seg, q, cq, known, driver, slots, 64);
    if (!expr) {
      if (!(q.min_should_match >= 2 && q.min_should_match < n_should))
        return fail_msg("a query that is a bitwise expression was handed to the scan", trial, (long)q.min_should_match);
      ++n_scan;
      continue;
    }
    if (known) {
      if (want != 0) return fail_msg("a query known to be empty has matches", trial, (long)want);
      ++n_known;
      continue;
    }
    std::vector<uint32_t> slot_term(slots.size());
    for (const auto &kv : slots) slot_term[kv.second] = kv.first;
    uint32_t got = 0;
    for (uint32_t w = 0; w < n_words; ++w) {  // count_bitmap_kernel's loop body
      uint32_t must = 0xFFFFFFFFu, nots = 0u, should = 0u, clause = 0u;
      for (uint32_t m = 0; m < cq.n_terms; ++m) {
        const uint32_t b = ((cq.narrow >> m) & 1u) ? bits[slot_term[(size_t)(uintptr_t)cq.dense[m]]][w] : cq.dense[m][w].x;
        const uint32_t kind = (cq.kinds >> (2u * m)) & 3u;
        if (kind == TQK_COUNT_MUST) {
          clause |= b;
          if ((cq.clause_end >> m) & 1u) {
            must &= clause;
            clause = 0u;
          }
        } else if (kind == TQK_COUNT_NOT) {
          nots |= b;
        } else {
          should |= b;
        }
      }
      uint32_t res = ((cq.flags & TQK_COUNT_HAS_MUST) ? must : should) & ~nots;
      if (cq.flags & TQK_COUNT_NEED_SHOULD) res &= should;
      got += (uint32_t)__builtin_popcount(res);
    }
    if (got != want) return fail_msg("expression count differs from the boolean semantics", (long)got, (long)want);
    ++n_expr;
  }
  printf("count: %u expressions, %u known empty, %u handed to the scan ok\n", n_expr, n_known, n_scan);
  return 0;
}

int main(int argc, char **argv) {
  const int seed = argc > 1 ? atoi(argv[1]) : 1;
  if (argc > 2 && !strcmp(argv[2], "share")) return check_share(seed);
  if (argc > 2 && !strcmp(argv[2], "pool")) return check_pool(seed);
  if (argc > 2 && !strcmp(argv[2], "dense")) return check_dense(seed);
  if (argc > 2 && !strcmp(argv[2], "ashare")) return check_ashare(seed);
  if (argc > 2 && !strcmp(argv[2], "bshare")) return check_bshare(seed);
  if (argc > 2 && !strcmp(argv[2], "count")) return check_count(seed);
  std::mt19937 rng(seed);
  auto uni = [&](uint32_t lo, uint32_t hi) { return std::uniform_int_distribution<uint32_t>(lo, hi)(rng); };
  PlanScratch ps;
  for (int shape = 0; shape < 3; ++shape) {  // 0 candidate unions, 1 AND-style, 2 OR windows
    Group &g = ps.groups[1];
    g.reset();
    g.mode = shape == 1 ? TQ_MODE_AND : TQ_MODE_OR;
    const uint32_t nq = shape == 0 ? 3000 : 400;
    for (uint32_t q = 0; q < nq; ++q) {
      TqdQuery dq{};
      dq.n_terms = uni(1, 6);
      dq.k = 10;
      dq.flags = uni(0, 3) ? TQD_QF_PRUNE : 0u;
      dq.tile_blocks = uni(1, 64);
      uint32_t at = 0;
      for (uint32_t i = 0; i < dq.n_terms; ++i) {
        dq.term[i] = uni(0, 300);
        dq.weight[i] = 20.0f / (float)(i + 1);
        dq.lead_tile_start[i] = at;
        at += uni(0, 9) == 0 ? 0u : uni(1, shape == 0 ? 4000 : 300);
      }
      for (uint32_t i = dq.n_terms; i <= TQ_MAX_TERMS; ++i) dq.lead_tile_start[i] = at;
      dq.n_lead = dq.n_terms;
      dq.n_tiles = uni(0, 19) == 0 ? 0u : at;  // some queries without tiles
      if (!dq.n_tiles)
        for (uint32_t i = 0; i <= TQ_MAX_TERMS; ++i) dq.lead_tile_start[i] = 0;
      g.queries.push_back(dq);
      g.tile_cost.push_back(uni(1, 200));
      g.out_index.push_back(q);
      g.max_k = 10;
    }
    if (build_group_chunks(g, shape == 2, ps) != TQ_OK) return fail_msg("build_group_chunks failed", shape);
    if (check(g, shape == 2)) return 1;
    printf("shape %d: %u queries, %u tiles, %u chunks ok\n", shape, nq, g.total_tiles, g.n_chunks);
  }
  return 0;
}
