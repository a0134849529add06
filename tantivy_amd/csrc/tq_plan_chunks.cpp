// tq_plan_chunks.cpp — the per-query kernels' launch tables: tiles -> chunks of about equal cost -> launch order
// (doc-range slices dealt over the XCDs), and the planner's thread pool size
// Part of the C ABI library of include/tantivy_amd.h (internal declarations: tq_internal.hpp).
#include "tq_internal.hpp"

void tq_free_plan_scratch(PlanScratch *p) { delete p; }

namespace tqi {

uint32_t plan_threads() {
  // (default 1 since round 4: with the per-query sorts gone and the intersections planned per leader the
  // calling thread plans a 10 000-query batch in about a millisecond; helper threads were no faster on
  // any bench workload and a descheduled helper — the GPU box shares its cores — stalled a batch for
  // up to 70 ms)
  static const uint32_t n = std::min<uint32_t>(16u, std::max<uint32_t>(1u, tune_u32("TQ_PLAN_THREADS", 1)));
  return n;
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
// batches below this many chunks are planned by the calling thread alone (TQ_PLAN_PAR_MIN: tests)
static const uint32_t kPlanParMin = tune_u32("TQ_PLAN_PAR_MIN", 16384);
static const uint32_t kOrChunkMul = std::max<uint32_t>(1u, tune_u32("TQ_OR_CHUNK_MUL", 4));
static const uint32_t kOrChunkMulSmallK = std::max<uint32_t>(1u, tune_u32("TQ_OR_CHUNK_MUL_SMALLK", 8));
// boolean queries (the union kernel's BOOL instantiation): measured on the bench shapes, kernel / host ms
// per 2000 queries: x8 6.05 / 4.26, x4 6.02 / 3.48, x2 6.04 / 1.77, x1 6.32 / 1.13 — the chunk records
// and partial lists of 1 M chunks bought nothing
static const uint32_t kBoolChunkMul = std::max<uint32_t>(1u, tune_u32("TQ_BOOL_CHUNK_MUL", 2));

// tiles -> chunks of one launch group: runs of consecutive tiles of about equal estimated cost,
// their launch order (doc-range slices) and the number of partial lists per query
int build_group_chunks(Group &g, bool or_windows, PlanScratch &ps, bool boolean_group) {
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
  // a chunk is at least 128 cost units (two 64-block tiles of an intersection) — unless the whole launch would then
  // be a few dozen wavefronts: a small batch (one query over a 50 k-doc leader = 4 chunks) is cut into about
  // kSmallChunks chunks of at least 8 units, tq_search.cpp sizes its tiles to match and the partial lists that makes
  // are merged in two levels
  // (only the smallest launches: from 16 queries on more wavefronts measured SLOWER — every wavefront starts from
  // the thresholds it finds, and a query cut into a hundred of them scores most of its matches before any of them
  // has a k-th best score to share: profiles/r05_small_batches.txt)
  static const uint64_t kSmallChunks = std::max<uint32_t>(1u, tune_u32("TQ_AND_SMALL_TILES", 2048));
  static const uint64_t kSmallCost = tune_u32("TQ_AND_SMALL_BLOCKS", 8192);
  const uint64_t cost_floor = or_win ? 1u : (total_cost < kSmallCost ? std::min<uint64_t>(128u, std::max<uint64_t>(8u, total_cost / kSmallChunks)) : 128u);
  const uint64_t cost_target = std::max<uint64_t>(cost_floor, (total_cost + n_target - 1) / n_target);
  pt("costs");
  // Every query with the same number of tiles of the same cost (the phrase sweep: a tile is 2 048 bitmap words whatever
  // the query): the records follow from arithmetic — chunk j of query q = tiles [j tpc, (j + 1) tpc), launched j-major
  // (all queries' first doc range, then the next) — instead of from the packing loop, the counting sort by slice and the
  // deal (0.72 ms per 1 000-phrase batch, which had become the step's bound once the kernels took 0.77 ms).
  if (!or_cand && !or_win && !g.queries.empty() && g.queries[0].n_tiles) {
    bool uniform = true;
    const uint32_t nt0 = g.queries[0].n_tiles, tc0 = std::max<uint32_t>(1u, g.tile_cost[0]);
    for (size_t i = 1; i < g.queries.size() && uniform; ++i)
      uniform = g.queries[i].n_tiles == nt0 && std::max<uint32_t>(1u, g.tile_cost[i]) == tc0;
    if (uniform && g.queries.size() >= 64) {
      const uint32_t tpc = (uint32_t)std::max<uint64_t>(1u, std::min<uint64_t>(nt0, (cost_target + tc0 - 1) / tc0));
      const uint32_t cpq = (nt0 + tpc - 1) / tpc;
      const size_t nq = g.queries.size();
      if ((uint64_t)cpq * nq > 0x7FFFFFFFull) return fail(TQ_ERR_UNSUPPORTED, "batch too large (chunks)");
      g.n_chunks = (uint32_t)(cpq * nq);
      g.chunk_recs.resize(g.n_chunks);
      for (size_t q = 0; q < nq; ++q) {
        TqdQuery &dq = g.queries[q];
        dq.part_start = 0;
        dq.chunk_first = (uint32_t)(q * cpq);
        dq.n_parts = cpq;
      }
      uint4 *out = g.chunk_recs.data();
      for (uint32_t j = 0; j < cpq; ++j) {
        const uint32_t t0 = j * tpc, t1 = std::min<uint32_t>(nt0, t0 + tpc);
        for (size_t q = 0; q < nq; ++q) {
          const uint32_t base = g.tile_starts[q];
          *out++ = make_uint4(base + t0, base + t1, (uint32_t)q, (uint32_t)(q * cpq + j));
        }
      }
      pt("uniform records");
      return TQ_OK;
    }
  }
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

}  // namespace tqi
