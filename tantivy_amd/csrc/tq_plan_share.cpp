// tq_plan_share.cpp — the term-major launches' planners: build_share_plan (pure unions: leads and tasks per term,
// tq_ushare.hip) and build_ashare_plan (intersections: leads per leader list, twins, warm-up and main tasks,
// tq_ashare.hip)
// Part of the C ABI library of include/tantivy_amd.h (internal declarations: tq_internal.hpp).
#include "tq_internal.hpp"

namespace tqi {

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
        {
          const TermHost &tm = s->terms[dq.term[m]];
          static const bool kUseRdir = tune_u32("TQ_US_RDIR", 1) != 0;
          if (s->opt.use_dense && tm.dense_blob) {
            ld.dense_off[m - li - 1u] = off_of(tm.dense_blob);
            ld.tf8_off[m - li - 1u] = off_of(tm.tf8_blob);
          } else if (kUseRdir && s->opt.use_dense && tm.rdir_blob && s->rdir_span_ok) {
            // a list without a bitmap but with a range directory (rdir_lookup, tq_common.hpp): directory | shift (the
            // tables' offsets are multiples of 32: low bits set = a directory), entries
            ld.dense_off[m - li - 1u] = off_of(tm.rdir_blob) | tm.rdir_shift;
            ld.tf8_off[m - li - 1u] = off_of(tm.rdir_ent);
          }
        }
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
  // (round 6, after the range directories: or5 4096 / 5120 / 6144 / 8192: 2.48 / 2.32 / 2.39 / 2.50 ms; mixed 2048 / 3072 /
  // 4096 / 6144: 6.74 / 5.63 / 5.41 / 5.59)
  const uint32_t kPhaseTasks = kPhaseTasksEnv ? kPhaseTasksEnv : (g.max_k <= 16u ? 4096u : 5120u);
  uint64_t phase_cost[TQD_US_MAX_TERMS] = {};
  for (size_t r0 = 0; r0 < keys.size();) {
    size_t r1 = r0;
    while (r1 < keys.size() && keys[r1].key == keys[r0].key && keys[r1].term == keys[r0].term) ++r1;
    const uint32_t n_run = (uint32_t)(r1 - r0);
    const uint32_t n_groups = (n_run + kGroupMax - 1) / kGroupMax;
    phase_cost[keys[r0].key >> 56] += (uint64_t)s->terms[keys[r0].term].n_blocks * (4u * n_groups + n_run);
    r0 = r1;
  }
  // runs of one (position, term, cache): groups of leads x runs of blocks.  Every (task, lead) pair may
  // append k entries to its query's result list: if the lists of the batch would not fit the budget
  // (TQ_US_LIST_MB; 10 000 five-term unions at k = 100 asked for several GB) the tasks are made longer —
  // fewer pairs, the same blocks — instead of failing the batch
  static const uint64_t kListBudget = (uint64_t)std::max<uint32_t>(1u, tune_u32("TQ_US_LIST_MB", 2048)) << 20;
  uint32_t stretch = 1;
retry_tasks:
  tasks.clear();
  pairs.assign(nq, 0u);
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
    bpt = (uint32_t)std::min<uint64_t>(0xFFFFu, (uint64_t)bpt * stretch);
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
  {
    uint64_t total = 0;
    for (size_t q = 0; q < nq; ++q) total += (uint64_t)pairs[q] * g.queries[q].k;
    if ((total * sizeof(uint64_t) > kListBudget || total > 0xFFFFFFFFull) && stretch < 65536u) {
      stretch *= 2u;
      goto retry_tasks;
    }
  }
  uint64_t entries = 0;
  for (size_t q = 0; q < nq; ++q) {
    TqdQuery &dq = g.queries[q];
    const uint64_t cap = (uint64_t)pairs[q] * dq.k;
    if (entries + cap > 0xFFFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (result lists)");
    dq.part_start = (uint32_t)entries;
    dq.n_parts = (uint32_t)cap;
    dq.chunk_first = (uint32_t)q;  // (merge_lists_kernel: the query whose list_count word counts this list)
    entries += cap;
  }
  g.list_entries = entries;
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
int build_ashare_plan(tq_segment *s, Group &g, PlanScratch &ps, bool boolean) {
  PlanScratch::ASharePlan &A = ps.ap[boolean ? 1 : 0];
  // (block, lead) pairs per task of the intersections: 0 = sized per batch so that the launch has about
  // kTargetTasks tasks (three per resident wavefront), between 64 and 512 pairs — with identical queries evaluated
  // once the headline batch holds 2.4 M pairs: at round 4's fixed 512 that was 5.7 k tasks for 8 192 resident
  // wavefronts (no balance, no threshold feedback between tasks: 1.23 ms; 128 pairs: 0.94 ms), while the batch
  // without repeats (6 M pairs) ran best at 256
  static const uint32_t kTaskPairsEnv = tune_u32("TQ_AS_TASK_PAIRS", 0);
  static const uint32_t kTargetTasks = std::max<uint32_t>(1u, tune_u32("TQ_AS_TARGET_TASKS", 24576));
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
  static const bool kUseRanges = tune_u32("TQ_AS_BOUND", 3) != 0;
  const size_t nq = g.queries.size();
  A.over_budget = false;
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
  std::vector<TqdALead> &leads = A.aleads, &unsorted = A.aleads_unsorted;
  std::vector<ALeadKey> &keys = A.alead_keys;
  // (leads per query: one for an intersection; one per list of the lead set for a boolean query)
  std::vector<uint32_t> &lead0 = A.alead_first;  // first lead of every query
  // Identical queries (lists, weights, k — and roles for boolean queries) are evaluated ONCE per batch: the
  // first of them (the smallest query index) gets leads, the others read its result list at merge time
  // (TqdQuery::chunk_first = the owner of the list, merge_lists_kernel).  Round 4 evaluated a repeated query
  // once per GROUP of 32 leads: the 267 copies of the headline batch's most frequent pair had its leader
  // decoded nine times.  The owners are found BEFORE any lead is built (an open-addressing table over the
  // query hashes: 6 of 10 headline queries never reach the sort).  TQ_AS_DEDUPE=0: one lead per query,
  // identical queries found as neighbours of the sorted table (twins inside a group).
  static const bool kDedupe = tune_u32("TQ_AS_DEDUPE", 1) != 0;
  std::vector<uint32_t> &owner = A.aowner;
  std::vector<uint64_t> &sigs = A.aq_sig;
  owner.resize(nq);
  sigs.resize(nq);
  // (what a query of <= 2 lists IS, in 24 bytes: the table compares these instead of two 328-byte descriptors)
  std::vector<PlanScratch::ASharePlan::QKey> &qk = A.aq_key;
  qk.resize(nq);
  for (size_t q = 0; q < nq; ++q) {
    const TqdQuery &dq = g.queries[q];
    uint64_t sig = 0x9E3779B97F4A7C15ull * (uint64_t)(dq.n_terms | (dq.k << 8));
    for (uint32_t m = 0; m < dq.n_terms; ++m) {
      uint32_t wb;
      memcpy(&wb, &dq.weight[m], sizeof wb);
      sig = (sig ^ (((uint64_t)dq.term[m] << 32) | wb)) * 0xFF51AFD7ED558CCDull;
      sig ^= sig >> 29;
    }
    if (boolean) {
      sig = (sig ^ (((uint64_t)dq.roles << 32) | dq.clause_end)) * 0xFF51AFD7ED558CCDull;
      sig = (sig ^ (((uint64_t)dq.n_lead << 40) | ((uint64_t)dq.n_opt_lead << 20) | dq.min_should)) * 0xFF51AFD7ED558CCDull;
    }
    sigs[q] = sig;
    owner[q] = (uint32_t)q;
    PlanScratch::ASharePlan::QKey &key = qk[q];
    key.t0 = dq.term[0];
    key.t1 = dq.n_terms > 1 ? dq.term[1] : 0u;
    memcpy(&key.w0, &dq.weight[0], 4);
    memcpy(&key.w1, &dq.weight[dq.n_terms > 1 ? 1 : 0], 4);
    key.k = dq.k;
    key.shape = dq.n_terms | (dq.cache_idx << 8) | (!boolean && dq.n_terms <= 2 ? 0x80000000u : 0u);  // bit 31: the key is the whole query
  }
  pt("sigs");
  if (kDedupe && nq > 1) {
    size_t cap = 16;
    while (cap < 2 * nq) cap <<= 1;
    std::vector<uint64_t> &slot = A.aq_slot;  // (32 bits of the hash) << 32 | query, or ~0
    slot.assign(cap, ~0ull);
    auto same = [&](size_t o, size_t q) -> bool {
      if (memcmp(&qk[o], &qk[q], sizeof(qk[0]))) return false;
      if (qk[q].shape >> 31) return true;
      const TqdQuery &a = g.queries[o], &b = g.queries[q];
      if (boolean && (a.roles != b.roles || a.clause_end != b.clause_end || a.n_lead != b.n_lead ||
                      a.n_opt_lead != b.n_opt_lead || a.min_should != b.min_should))
        return false;
      return !memcmp(a.term, b.term, a.n_terms * sizeof(uint32_t)) && !memcmp(a.weight, b.weight, a.n_terms * sizeof(float));
    };
    constexpr size_t kAhead = 8;
    for (size_t q = 0; q < nq; ++q) {
      if (q + kAhead < nq) __builtin_prefetch(&slot[(size_t)(sigs[q + kAhead] >> 20) & (cap - 1)], 1);
      const uint64_t tag = sigs[q] & 0xFFFFFFFF00000000ull;
      size_t h = (size_t)(sigs[q] >> 20) & (cap - 1);
      for (;; h = (h + 1) & (cap - 1)) {
        const uint64_t e = slot[h];
        if (e == ~0ull) {
          slot[h] = tag | (uint64_t)q;
          break;
        }
        if ((e & 0xFFFFFFFF00000000ull) == tag && same((size_t)(uint32_t)e, q)) {  // (the hash only proposes)
          owner[q] = (uint32_t)e;
          break;
        }
      }
    }
  }
  pt("dedupe");
  lead0.resize(nq + 1);
  size_t nl = 0;
  for (size_t q = 0; q < nq; ++q) {
    lead0[q] = (uint32_t)nl;
    if (owner[q] != q) continue;
    nl += boolean ? std::max<uint32_t>(1u, g.queries[q].n_lead) : 1u;
  }
  lead0[nq] = (uint32_t)nl;
  leads.resize(nl);
  unsorted.resize(nl);
  keys.resize(nl);
  if (boolean) A.alists.assign(nq * TQD_AS_MAX_TERMS, make_uint2(0u, 0u));
  A.any_rdir = false;
  std::atomic<bool> any_rdir{false};
  const uint32_t fill_slabs = nq >= 4096 ? plan_threads() : 1u;
  parallel_slabs(fill_slabs, [&](uint32_t sb) {
    const size_t q0 = nq * sb / fill_slabs, q1 = nq * (sb + 1) / fill_slabs;
    for (size_t q = q0; q < q1; ++q) {
      if (owner[q] != q) continue;
      const TqdQuery &dq = g.queries[q];
      const uint64_t sig = sigs[q];
      auto bit_of = [&](uint32_t handle) -> uint32_t {  // the list's doc-matrix bit: column (exact) or signature (maybe)
        const uint32_t col = column_of(handle);
        const uint32_t sig1 = !col ? (s->h_dterms[handle].has_freq >> 16) & 0xFFu : 0u;
        return col ? col : (sig1 ? TQD_SIG_SHIFT + (sig1 - 1u) : 0u);
      };
      if (!boolean) {
        TqdALead ld{};
        ld.query = (uint32_t)q;
        ld.w = dq.weight[0];
        float rest = 0.0f;
        uint64_t mask = 0;
        for (uint32_t m = 1; m < dq.n_terms; ++m) {
          rest += dq.weight[m];
          const uint32_t bitpos = bit_of(dq.term[m]);
          if (bitpos) mask |= 1ull << bitpos;
        }
        ld.rest = rest;
        ld.mask_lo = (uint32_t)mask;
        ld.mask_hi = (uint32_t)(mask >> 32);
        ld.info = dq.n_terms | (column_of(dq.term[1]) ? 0x100u : 0u);
        {  // list 1's tf classes in the segment's class matrix (bits 10-15: column slot + 1): its tf without bitmap word + tf byte
          static const bool kUseCls = tune_u32("TQ_AS_CLS", 1) != 0;
          const uint32_t hf = s->h_dterms[dq.term[1]].has_freq;
          if (kUseCls && s->d_doccls && ((hf >> 24) & 1u)) ld.info |= ((hf >> 8) & 0x3Fu) << 10;
        }
        {  // (list 1's own tables, or the ones built for probes: build_probe_tables)
          const TermHost &t1 = s->terms[dq.term[1]];
          const bool own = t1.dense_blob && t1.tf8_blob;
          static const bool kUseRdir = tune_u32("TQ_AS_RDIR", 1) != 0;
          if (!own && dq.n_terms == 2u && t1.rdir_blob && s->rdir_span_ok && (kUseRdir || !(t1.probe_dense_blob && t1.probe_tf8_blob))) {
            // its range directory: dense_off = the directory, tf8_off = the entries, info bits 16-20 = the shift
            ld.info |= TQD_AL_RDIR | (t1.rdir_shift << 16);
            ld.dense_off = off_of(t1.rdir_blob);
            ld.tf8_off = off_of(t1.rdir_ent);
          } else {
            ld.dense_off = off_of(own ? t1.dense_blob : t1.probe_dense_blob);
            ld.tf8_off = off_of(own ? t1.tf8_blob : t1.probe_tf8_blob);
          }
        }
        {  // list 1's range maxima (tq_terms.cpp build_rmax), the split of `rest` the kernel bounds with them
          const TermHost &t1 = s->terms[dq.term[1]];
          const float w1 = dq.weight[1];
          float others = 0.0f;
          for (uint32_t m = 2; m < dq.n_terms; ++m) others += dq.weight[m];
          // (a list probed through its range directory: the directory says which leader blocks it has postings in —
          // the kernel reads it through dense_off — and the list's largest tf/(tf + norm) stands in for range maxima)
          const bool rdir = (ld.info & TQD_AL_RDIR) != 0u;
          ld.excl_lo = kUseRanges && !rdir && t1.rmax_blob ? off_of(t1.rmax_blob) : 0u;
          memcpy(&ld.excl_hi, &w1, sizeof(float));
          memcpy(&ld.any1_lo, &others, sizeof(float));
          ld.any1_hi = (ld.excl_lo || (kUseRanges && rdir)) ? t1.rmax_list : 255u;
        }
        ld.k = dq.k;
        ld.thr_row = dq.thr_index;
        unsorted[lead0[q]] = ld;
        keys[lead0[q]] = ALeadKey{((uint64_t)dq.term[0] << 8) | (uint64_t)(dq.cache_idx & 0xFFu), mask, sig, lead0[q], 0u};
        continue;
      }
      // boolean query (plan_bool_query's layout: [optional leading Should lists | lead Must clause | other
      // Must clauses | MustNot lists | Should lists]): one lead per list of the lead set
      const uint32_t n_lead = std::max<uint32_t>(1u, dq.n_lead);
      for (uint32_t m = 0; m < dq.n_terms; ++m)
      {  // (the list's own tables, or the ones built for boolean probes: build_probe_tables)
        const TermHost &th = s->terms[dq.term[m]];
        const bool own = th.dense_blob && th.tf8_blob;
        if (!own && !(th.probe_dense_blob && th.probe_tf8_blob) && th.rdir_blob && s->rdir_span_ok)  // (its range directory | shift, its entries)
        {
          A.alists[q * TQD_AS_MAX_TERMS + m] = make_uint2(off_of(th.rdir_blob) | th.rdir_shift, off_of(th.rdir_ent));
          any_rdir.store(true, std::memory_order_relaxed);
        } else
          A.alists[q * TQD_AS_MAX_TERMS + m] = make_uint2(off_of(own ? th.dense_blob : th.probe_dense_blob), off_of(own ? th.tf8_blob : th.probe_tf8_blob));
      }
      for (uint32_t li = 0; li < n_lead; ++li) {
        TqdALead ld{};
        ld.query = (uint32_t)q;
        ld.w = dq.weight[li];
        float rest = 0.0f;
        for (uint32_t m = dq.n_terms; m-- > li + 1u;) rest += dq.weight[m];  // (suffix[li + 1], summed as tq_union.hip does)
        uint64_t excl = 0, any[TQ_MAX_TERMS + 1] = {0};
        uint32_t n_any = 0;
        if (li < dq.n_opt_lead) {  // an optional list leads: the doc must be in the lead Must clause as well
          uint64_t m1 = 0;
          for (uint32_t m = dq.n_opt_lead; m < dq.n_lead; ++m) {
            const uint32_t b = bit_of(dq.term[m]);
            m1 = b && m1 != ~0ull ? m1 | (1ull << b) : ~0ull;
          }
          if (m1 != ~0ull && m1) any[n_any++] = m1;
        }
        uint64_t cm = 0;
        for (uint32_t m = 0; m < dq.n_terms; ++m) {
          if (m == li) continue;
          const uint32_t role = (dq.roles >> (2u * m)) & 3u;
          const uint32_t col = column_of(dq.term[m]);
          if (role == TQD_ROLE_MUST_NOT) {
            if (col) excl |= 1ull << col;
          } else if (m < dq.n_lead) {
            if (m < li && col) excl |= 1ull << col;  // held by an earlier list of the lead set: that lead's doc
          } else if (role == TQD_ROLE_MUST) {
            const uint32_t b = bit_of(dq.term[m]);
            cm = b && cm != ~0ull ? cm | (1ull << b) : ~0ull;
            if ((dq.clause_end >> m) & 1u) {
              if (cm != ~0ull && cm) any[n_any++] = cm;
              cm = 0;
            }
          }
        }
        // one byte per list after the leader that can add to the score and has a doc-matrix bit: a doc
        // without the bit does not get the list's weight in its bound, and the scoring stage does not probe
        // the list; the lists without a bit add `rest_base`.  Lists the exclusion mask has ruled out (columns
        // of MustNot lists and of the lead-set lists before the leader) are never probed (info bits 20-27).
        uint64_t bytes = 0;
        float rest_base = 0.0f;
        uint32_t maybe = 0xFFu;
        for (uint32_t m = 0; m < dq.n_terms; ++m) {
          if (m == li) continue;
          const uint32_t role = (dq.roles >> (2u * m)) & 3u;
          if ((role == TQD_ROLE_MUST_NOT || m < li) && column_of(dq.term[m])) maybe &= ~(1u << m);
          if (m < li || role == TQD_ROLE_MUST_NOT) continue;
          const uint32_t b = bit_of(dq.term[m]);
          bytes |= (uint64_t)b << (8u * m);
          if (!b) rest_base += dq.weight[m];
        }
        ld.dense_off = (uint32_t)bytes;
        ld.tf8_off = (uint32_t)(bytes >> 32);
        memcpy(&ld.mask_lo, &rest_base, sizeof(float));
        ld.rest = rest;
        ld.excl_lo = (uint32_t)excl;
        ld.excl_hi = (uint32_t)(excl >> 32);
        if (n_any > 0) {
          ld.any1_lo = (uint32_t)any[0];
          ld.any1_hi = (uint32_t)(any[0] >> 32);
        }
        if (n_any > 1) {
          ld.any2_lo = (uint32_t)any[1];
          ld.any2_hi = (uint32_t)(any[1] >> 32);
        }
        ld.info = dq.n_terms | (li << 16) | (maybe << 20);
        ld.k = dq.k;
        ld.thr_row = dq.thr_index;
        const size_t L = lead0[q] + li;
        unsorted[L] = ld;
        // (sort key: the masks that decide membership, folded; the hash carries the leading list too)
        const uint64_t mk = excl ^ (n_any > 0 ? any[0] * 0x9E3779B97F4A7C15ull : 0ull) ^ (n_any > 1 ? any[1] * 0xC2B2AE3D27D4EB4Full : 0ull);
        keys[L] = ALeadKey{((uint64_t)dq.term[li] << 8) | (uint64_t)(dq.cache_idx & 0xFFu), mk,
                           (sig ^ li) * 0xFF51AFD7ED558CCDull, (uint32_t)L, 0u};
      }
    }
  });
  A.any_rdir = any_rdir.load(std::memory_order_relaxed);
  pt("leads");
  // ---- order: (leader, cache), then mask (the kernel keeps a block's membership ballots across consecutive
  // leads with the same mask), stable in the query index: two counting sorts, 11 bits of a hash of the mask,
  // then the leader (a comparison sort of the table was 0.40 of this function's 1.08 ms on the headline
  // batch).  TQ_AS_DEDUPE=0 needs identical queries as neighbours: buckets by leader, then every bucket
  // sorted by (mask, query hash).
  {
    std::vector<uint32_t> &cnt = A.alead_bucket;
    const size_t nt = s->terms.size();
    std::vector<ALeadKey> &tmp = A.alead_keys2;
    tmp.resize(nl);
    std::vector<uint32_t> &at = A.alead_bucket_at;
    if (kDedupe) {
      at.assign(kALeadMaskBins + 1, 0u);
      uint64_t cache_diff = 0;
      for (size_t i = 0; i < nl; ++i) {
        ++at[alead_mask_bin(keys[i].mask) + 1];
        cache_diff |= (keys[i].k1 ^ keys[0].k1) & 0xFFu;
      }
      for (uint32_t b = 0; b < kALeadMaskBins; ++b) at[b + 1] += at[b];
      for (size_t i = 0; i < nl; ++i) tmp[at[alead_mask_bin(keys[i].mask)]++] = keys[i];
      keys.swap(tmp);
      if (cache_diff) {  // more than one Bm25 cache in the batch (fields, boosts of the average): (leader, cache) runs
        at.assign(257, 0u);
        for (size_t i = 0; i < nl; ++i) ++at[(keys[i].k1 & 0xFFu) + 1];
        for (uint32_t b = 0; b < 256; ++b) at[b + 1] += at[b];
        for (size_t i = 0; i < nl; ++i) tmp[at[keys[i].k1 & 0xFFu]++] = keys[i];
        keys.swap(tmp);
      }
    }
    cnt.assign(nt + 1, 0u);
    for (size_t q = 0; q < nl; ++q) ++cnt[(size_t)(keys[q].k1 >> 8) + 1];
    for (size_t t = 0; t < nt; ++t) cnt[t + 1] += cnt[t];
    at.assign(cnt.begin(), cnt.end() - 1);
    for (size_t q = 0; q < nl; ++q) tmp[at[(size_t)(keys[q].k1 >> 8)]++] = keys[q];
    keys.swap(tmp);
    if (!kDedupe) {
      // non-empty buckets, cut into slabs of about equal size
      std::vector<uint32_t> &starts = A.alead_bucket_starts;
      starts.clear();
      for (size_t t = 0; t < nt; ++t)
        if (cnt[t + 1] > cnt[t]) starts.push_back(cnt[t]);
      starts.push_back((uint32_t)nl);
      const uint32_t n_b = (uint32_t)starts.size() - 1u;
      const uint32_t sort_slabs = nl >= 4096 ? std::min<uint32_t>(plan_threads(), std::max<uint32_t>(1u, n_b)) : 1u;
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
  }
  pt("sort");
  auto same_query = [&](const ALeadKey &a, const ALeadKey &b) -> bool {  // (the hash only proposes)
    if (a.k1 != b.k1 || a.mask != b.mask || a.sig != b.sig) return false;
    const TqdALead &la = unsorted[a.q], &lb = unsorted[b.q];
    if (la.info != lb.info || la.k != lb.k || memcmp(&la.w, &lb.w, 4) || memcmp(&la.rest, &lb.rest, 4) ||
        la.dense_off != lb.dense_off)
      return false;
    if (!boolean && (la.info & 31u) == 2u) return true;  // (leader, list 1 — every list has its own bitmap —, both weights, k)
    const TqdQuery &qa = g.queries[la.query], &qb = g.queries[lb.query];
    if (boolean && (qa.roles != qb.roles || qa.clause_end != qb.clause_end || qa.n_lead != qb.n_lead ||
                    qa.n_opt_lead != qb.n_opt_lead || qa.min_should != qb.min_should))
      return false;
    return !memcmp(qa.term, qb.term, qa.n_terms * sizeof(uint32_t)) &&
           !memcmp(qa.weight, qb.weight, qa.n_terms * sizeof(float));
  };
  // TQ_AS_DEDUPE=0: identical queries share one row of threshold slots, whatever groups they end up in (a slot
  // is hash(doc): the same doc lands in the same slot whichever group scored it)
  std::vector<uint8_t> &same_as_prev = A.alead_same;
  same_as_prev.resize(nl);
  const uint32_t gather_slabs = nl >= 4096 ? plan_threads() : 1u;
  parallel_slabs(gather_slabs, [&](uint32_t sb) {
    const size_t i0 = nl * sb / gather_slabs, i1 = nl * (sb + 1) / gather_slabs;
    for (size_t i = i0; i < i1; ++i) {
      leads[i] = unsorted[keys[i].q];
      same_as_prev[i] = !kDedupe && i && same_query(keys[i], keys[i - 1]) ? 1 : 0;
    }
  });
  if (!kDedupe)
    for (size_t i = 1; i < nl; ++i)
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
  std::vector<uint4> &tasks = A.atasks;
  std::vector<uint32_t> &pairs = A.apairs;
  pairs.resize(nq);
  // the runs of one (leader, cache): their groups, task sizes and where their tasks start
  std::vector<PlanScratch::ASharePlan::ARun> &runs = A.aruns;
  // (boolean leads: 128 pairs per task measured 8 % faster than 512, 64 pairs 10 % slower — their scoring stage is long, shorter
  // tasks balance the tail; intersections: 512, §3.1a)
  static const uint32_t kBoolTaskPairs = std::max<uint32_t>(32u, tune_u32("TQ_BS_TASK_PAIRS", 128));
  uint32_t task_pairs = boolean ? kBoolTaskPairs : kTaskPairsEnv;
  if (!boolean && !task_pairs) {
    uint64_t total_pairs = 0;
    for (size_t r0 = 0; r0 < nl;) {
      size_t r1 = r0;
      while (r1 < nl && keys[r1].k1 == keys[r0].k1) ++r1;
      total_pairs += (uint64_t)(r1 - r0) * s->terms[(size_t)(keys[r0].k1 >> 8)].n_blocks;
      r0 = r1;
    }
    static const uint32_t kTaskPairsMin = std::max<uint32_t>(4u, tune_u32("TQ_AS_TASK_PAIRS_MIN", 64));
    task_pairs = (uint32_t)std::min<uint64_t>(512u, std::max<uint64_t>(kTaskPairsMin, total_pairs / kTargetTasks));
  }
  task_pairs = std::max<uint32_t>(4u, task_pairs);
  // `stretch` multiplies a run's blocks per task AFTER the cap of kTaskBlocksMax (as build_share_plan does):
  // doubling the pairs per task stopped shrinking the lists once every run sat at the cap, and a large-k batch
  // over long leaders went on to allocate gigabytes of result lists.  The kernel walks a task in
  // tiles of TQD_AS_TILE blocks, so long tasks are safe.
  // (a lead's docs scored per task are counted in 16 bits next to its list length: 511 blocks x 128 docs fit)
  const uint32_t kStretchedBlocksMax = std::max<uint32_t>(kTaskBlocksMax, 511u);
  uint32_t stretch = 1;
  size_t n_tasks = 0;
  for (;;) {
    runs.clear();
    n_tasks = 0;
    uint64_t entries = 0;
    bool can_stretch = false;  // some run still has more than one main task
    for (size_t r0 = 0; r0 < nl;) {
      size_t r1 = r0;
      uint64_t k_sum = 0;
      while (r1 < nl && keys[r1].k1 == keys[r0].k1) k_sum += leads[r1++].k;
      PlanScratch::ASharePlan::ARun R;
      R.r0 = (uint32_t)r0;
      R.r1 = (uint32_t)r1;
      R.term = (uint32_t)(keys[r0].k1 >> 8);
      R.cache = (uint32_t)keys[r0].k1 & 0xFFu;
      R.n_blocks = s->terms[R.term].n_blocks;
      const uint32_t n_run = (uint32_t)(r1 - r0);
      const uint32_t n_groups = (n_run + kGroupMax - 1) / kGroupMax;
      R.per_group = (n_run + n_groups - 1) / n_groups;
      R.n_groups = (n_run + R.per_group - 1) / R.per_group;  // (the non-empty ones)
      R.bpt = (uint32_t)std::min<uint64_t>(kStretchedBlocksMax, (uint64_t)std::min<uint32_t>(kTaskBlocksMax, std::max<uint32_t>(1u, task_pairs / R.per_group)) * stretch);
      R.nb_warm = (uint32_t)((uint64_t)R.n_blocks * kWarmPermille / 1000u);
      R.n_runs = (R.nb_warm + kWarmBlocks - 1) / kWarmBlocks + (R.n_blocks - R.nb_warm + R.bpt - 1) / R.bpt;
      can_stretch = can_stretch || (R.bpt < kStretchedBlocksMax && R.bpt < R.n_blocks);
      R.task0 = n_tasks;
      n_tasks += (size_t)R.n_runs * R.n_groups;
      entries += (uint64_t)R.n_runs * k_sum;
      runs.push_back(R);
      r0 = r1;
    }
    if (entries * sizeof(uint64_t) <= kListBudget && entries <= 0xFFFFFFFFull) break;
    if (!can_stretch) {  // the longest tasks the kernel takes, and the lists still do not fit: the caller plans
      A.over_budget = true;  // the batch again without the shared launch (search_batch_impl)
      return fail(TQ_ERR_UNSUPPORTED, "batch too large (result lists of the shared launch: %llu entries over the %llu MB budget)",
                  (unsigned long long)entries, (unsigned long long)(kListBudget >> 20));
    }
    stretch *= 2u;
  }
  if (n_tasks > 0x7FFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (tasks)");
  tasks.resize(n_tasks);
  // slabs of runs of about equal task counts: each counts its tasks by doc slice; the (slice, slab) prefix
  // sums give every slab its places in the launch order (a stable counting sort: slice 0 = the warm-up
  // launch, then the main launch's tasks by doc slice); a second walk over the runs writes the tasks there
  constexpr uint32_t kSl = 4098;
  const uint32_t t_slabs = n_tasks >= 16384 ? std::min<uint32_t>(plan_threads(), (uint32_t)runs.size()) : 1u;
  std::vector<uint32_t> &hist = A.atask_hist;
  hist.assign((size_t)t_slabs * kSl, 0u);
  std::vector<uint32_t> &slab_run = A.atask_slab_run;
  slab_run.assign(t_slabs + 1, (uint32_t)runs.size());
  {
    uint32_t sb = 0;
    slab_run[0] = 0;
    for (uint32_t r = 0; r < runs.size() && sb + 1 < t_slabs; ++r)
      if (runs[r].task0 >= n_tasks * (sb + 1) / t_slabs) slab_run[++sb] = r;
    for (uint32_t x = sb + 1; x < t_slabs; ++x) slab_run[x] = (uint32_t)runs.size();
  }
  auto walk_run = [&](const PlanScratch::ASharePlan::ARun &R, auto &&emit) {  // emit(j0, nb, slice) per run of blocks
    const uint64_t slice_mul = ((uint64_t)1 << 44) / R.n_blocks;  // (j0 << 12) / n_blocks without the division
    for (uint32_t j0 = 0; j0 < R.n_blocks;) {
      const bool warm = j0 < R.nb_warm;
      const uint32_t nb = warm ? std::min<uint32_t>(kWarmBlocks, R.nb_warm - j0) : std::min<uint32_t>(R.bpt, R.n_blocks - j0);
      emit(j0, nb, warm ? 0u : 1u + std::min<uint32_t>(4095u, (uint32_t)((j0 * slice_mul) >> 32)));
      j0 += nb;
    }
  };
  parallel_slabs(t_slabs, [&](uint32_t sb) {
    uint32_t *h = hist.data() + (size_t)sb * kSl;
    for (uint32_t r = slab_run[sb]; r < slab_run[sb + 1]; ++r) {
      const PlanScratch::ASharePlan::ARun &R = runs[r];
      for (uint32_t a = R.r0; a < R.r1; ++a) {  // twins: the same query as the lead before, inside one group
        const bool twin = (a - R.r0) % R.per_group != 0 && same_as_prev[a];
        leads[a].info = (leads[a].info & ~0x200u) | (twin ? 0x200u : 0u);
      }
      walk_run(R, [&](uint32_t, uint32_t, uint32_t slice) { h[slice] += R.n_groups; });
    }
  });
  {
    uint32_t run = 0;
    for (uint32_t sl = 0; sl < kSl; ++sl)
      for (uint32_t sb = 0; sb < t_slabs; ++sb) {
        const uint32_t n = hist[(size_t)sb * kSl + sl];
        hist[(size_t)sb * kSl + sl] = run;  // becomes the slab's write position in this slice
        run += n;
        if (sl == 0 && sb + 1 == t_slabs) A.a_warm_tasks = run;
      }
  }
  parallel_slabs(t_slabs, [&](uint32_t sb) {
    uint32_t *h = hist.data() + (size_t)sb * kSl;
    for (uint32_t r = slab_run[sb]; r < slab_run[sb + 1]; ++r) {
      const PlanScratch::ASharePlan::ARun &R = runs[r];
      const uint32_t n_run = R.r1 - R.r0;
      walk_run(R, [&](uint32_t j0, uint32_t nb, uint32_t slice) {
        uint4 *out = tasks.data() + h[slice];
        h[slice] += R.n_groups;
        for (uint32_t gr = 0; gr < R.n_groups; ++gr) {
          const uint32_t l0 = gr * R.per_group, l1 = std::min<uint32_t>(n_run, l0 + R.per_group);
          out[gr] = make_uint4(R.term, j0, nb | ((l1 - l0) << 16) | (R.cache << 24), R.r0 + l0);
        }
      });
    }
  });
  pt("tasks");
  // result lists: k entries per (task, lead) pair of the query
  std::fill(pairs.begin(), pairs.end(), 0u);
  for (const PlanScratch::ASharePlan::ARun &R : runs)
    for (uint32_t a = R.r0; a < R.r1; ++a) pairs[leads[a].query] += R.n_runs;
  uint64_t entries = 0;
  for (size_t q = 0; q < nq; ++q) {
    TqdQuery &dq = g.queries[q];
    if (owner[q] != q) {  // reads the list of the identical query before it (owner[q] < q: already placed)
      dq.part_start = g.queries[owner[q]].part_start;
      dq.n_parts = g.queries[owner[q]].n_parts;
      dq.chunk_first = owner[q];
      continue;
    }
    const uint64_t cap = (uint64_t)pairs[q] * dq.k;
    if (entries + cap > 0xFFFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (result lists)");
    dq.part_start = (uint32_t)entries;
    dq.n_parts = (uint32_t)cap;
    dq.chunk_first = (uint32_t)q;  // (merge_lists_kernel: the query whose list_count word counts this list)
    entries += cap;
  }
  g.list_entries = entries;
  g.total_tiles = (uint32_t)tasks.size();
  g.n_chunks = (uint32_t)tasks.size();
  return TQ_OK;
}

}  // namespace tqi
