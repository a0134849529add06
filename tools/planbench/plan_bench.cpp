// Host-only timing of the chunk planner (build_group_chunks of tq_api.cpp) on a synthetic launch
// group shaped like the union half of the mixed bench stream: 5000 5-term candidate unions over a
// 256-term Zipf vocabulary, 10M docs.  No GPU needed: the planner is plain host code.
//   hipcc -O3 -std=c++17 -I tantivy_amd/csrc tools/planbench/plan_bench.cpp -o /tmp/plan_bench
#include "../../tantivy_amd/csrc/tq_internal.hpp"

#include <chrono>
#include <random>

// build_share_plan on the union half of the mixed stream: 5000 5-term unions, k = 10, 256 Zipf lists
// (the 145 densest with a bitmap, 40 of them with a doc-matrix column), 10M docs
static int bench_share(int n_queries, int reps) {
  const uint32_t max_doc = 10000000u, n_terms = 256;
  tq_segment seg;
  static uint8_t arena[1 << 21];
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    th.doc_freq = max_doc / 2 / (t + 1);
    th.n_blocks = (th.doc_freq + 127) / 128;
    TqdTerm dt{};
    dt.has_freq = 1u;
    if (t < 145) {
      th.dense_blob = arena + 4096u * t + 8u;
      th.tf8_blob = arena + 4096u * t + 2048u;
      if (t < TQD_MAT_SLOTS) dt.has_freq |= (t + 1u) << 8;
    } else {
      dt.has_freq |= ((t * 7u) % TQD_SIG_BITS + 1u) << 16;
    }
    seg.terms.push_back(th);
    seg.h_dterms.push_back(dt);
  }
  seg.max_doc = max_doc;
  seg.share_table_lo = (uint64_t)arena;
  std::mt19937 rng(7);
  std::vector<double> cdf(n_terms);
  double acc = 0;
  for (uint32_t r = 0; r < n_terms; ++r) cdf[r] = (acc += 1.0 / (r + 1));
  PlanScratch ps;
  double best = 1e9;
  for (int rep = 0; rep < reps; ++rep) {
    Group &g = ps.groups[5];
    g.reset();
    g.mode = TQ_MODE_OR;
    rng.seed(7);
    for (int q = 0; q < n_queries; ++q) {
      TqdQuery dq{};
      uint32_t picked[5], n = 0;
      while (n < 5) {
        const double u = std::uniform_real_distribution<double>(0, acc)(rng);
        const uint32_t r = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
        bool dup = false;
        for (uint32_t i = 0; i < n; ++i) dup |= picked[i] == r;
        if (!dup) picked[n++] = r;
      }
      std::sort(picked, picked + 5, [](uint32_t a, uint32_t b) { return a > b; });
      dq.n_terms = 5;
      dq.k = 10;
      dq.flags = TQD_QF_PRUNE;
      dq.thr_index = 4u * q;
      for (uint32_t i = 0; i < 5; ++i) {
        dq.term[i] = picked[i];
        dq.weight[i] = 30.0f / (float)(i + 1);
      }
      g.queries.push_back(dq);
      g.tile_cost.push_back(1);
      g.out_index.push_back((uint32_t)q);
      g.max_k = 10;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (build_share_plan(&seg, g, ps) != TQ_OK) return 1;
    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  printf("share: queries %d leads %zu tasks %zu: build_share_plan %.2f ms (best of %d)\n", n_queries,
         ps.leads.size(), ps.tasks.size(), best, reps);
  return 0;
}

// build_group_chunks on the headline batch: 2-term ANDs over the 256 Zipf lists, leader = the rarer list,
// 64-block tiles (all-dense group: tile cost = tile blocks)
static int bench_and(int n_queries, int reps) {
  const uint32_t max_doc = 10000000u, n_terms = 256;
  std::mt19937 rng(7);
  std::vector<double> cdf(n_terms);
  double acc = 0;
  for (uint32_t r = 0; r < n_terms; ++r) cdf[r] = (acc += 1.0 / (r + 1));
  PlanScratch ps;
  double best = 1e9;
  for (int rep = 0; rep < reps; ++rep) {
    Group &g = ps.groups[0];
    g.reset();
    g.mode = TQ_MODE_AND;
    rng.seed(7);
    for (int q = 0; q < n_queries; ++q) {
      TqdQuery dq{};
      uint32_t a, b;
      do {
        a = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), std::uniform_real_distribution<double>(0, acc)(rng)) - cdf.begin());
        b = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), std::uniform_real_distribution<double>(0, acc)(rng)) - cdf.begin());
      } while (a == b);
      const uint32_t lead = std::max(a, b);
      dq.n_terms = 2;
      dq.k = 10;
      dq.flags = TQD_QF_PRUNE;
      dq.term[0] = lead;
      dq.term[1] = std::min(a, b);
      dq.tile_blocks = TQD_AND_TILE;
      const uint32_t lead_blocks = (max_doc / 2 / (lead + 1) + 127) / 128;
      dq.n_tiles = (lead_blocks + dq.tile_blocks - 1) / dq.tile_blocks;
      g.queries.push_back(dq);
      g.tile_cost.push_back(dq.tile_blocks);
      g.out_index.push_back((uint32_t)q);
      g.max_k = 10;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (build_group_chunks(g, false, ps) != TQ_OK) return 1;
    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  printf("and: queries %d chunks %u tiles %u: build_group_chunks %.2f ms (best of %d)\n", n_queries,
         ps.groups[0].n_chunks, ps.groups[0].total_tiles, best, reps);
  return 0;
}

// build_ashare_plan on the headline batch: 2-term ANDs over the 256 Zipf lists whose other list has a
// bitmap (ranks < 64), leader = the rarer list
static int bench_ashare(int n_queries, int reps) {
  const uint32_t max_doc = 10000000u, n_terms = 256;
  tq_segment seg;
  static uint8_t arena[1 << 21];
  for (uint32_t t = 0; t < n_terms; ++t) {
    TermHost th;
    th.doc_freq = max_doc / 2 / (t + 1);
    th.n_blocks = (th.doc_freq + 127) / 128;
    TqdTerm dt{};
    dt.has_freq = 1u;
    if (t < 64) {
      th.dense_blob = arena + 4096u * t + 8u;
      th.tf8_blob = arena + 4096u * t + 2048u;
    }
    if (t < TQD_MAT_SLOTS) dt.has_freq |= (t + 1u) << 8;
    else dt.has_freq |= ((t * 7u) % TQD_SIG_BITS + 1u) << 16;
    seg.terms.push_back(th);
    seg.h_dterms.push_back(dt);
  }
  seg.max_doc = max_doc;
  seg.share_table_lo = (uint64_t)arena;
  std::mt19937 rng(7);
  std::vector<double> cdf(n_terms);
  double acc = 0;
  for (uint32_t r = 0; r < n_terms; ++r) cdf[r] = (acc += 1.0 / (r + 1));
  PlanScratch ps;
  double best = 1e9;
  std::vector<double> all_ms;
  for (int rep = 0; rep < reps; ++rep) {
    Group &g = ps.groups[8];
    g.reset();
    g.mode = TQ_MODE_AND;
    rng.seed(7);
    for (int q = 0; q < n_queries; ++q) {
      TqdQuery dq{};
      uint32_t a, b;
      do {
        a = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), std::uniform_real_distribution<double>(0, acc)(rng)) - cdf.begin());
        b = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), std::uniform_real_distribution<double>(0, acc)(rng)) - cdf.begin());
      } while (a == b || std::min(a, b) >= 64);
      dq.n_terms = 2;
      dq.k = 10;
      dq.flags = TQD_QF_PRUNE;
      dq.thr_index = (uint32_t)q;
      dq.term[0] = std::max(a, b);
      dq.term[1] = std::min(a, b);
      dq.weight[0] = 2.2f * logf(1.0f + (max_doc - seg.terms[dq.term[0]].doc_freq + 0.5f) / (seg.terms[dq.term[0]].doc_freq + 0.5f));
      dq.weight[1] = 2.2f * logf(1.0f + (max_doc - seg.terms[dq.term[1]].doc_freq + 0.5f) / (seg.terms[dq.term[1]].doc_freq + 0.5f));
      g.queries.push_back(dq);
      g.tile_cost.push_back(1);
      g.out_index.push_back((uint32_t)q);
      g.max_k = 10;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (build_ashare_plan(&seg, g, ps) != TQ_OK) return 1;
    all_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    best = std::min(best, all_ms.back());
  }
  std::sort(all_ms.begin(), all_ms.end());
  printf("ashare: queries %d tasks %zu (warm %u): build_ashare_plan %.2f ms (best of %d), median %.2f ms, p90 %.2f ms\n",
         n_queries, ps.ap[0].atasks.size(), ps.ap[0].a_warm_tasks, best, reps, all_ms[all_ms.size() / 2],
         all_ms[std::min(all_ms.size() - 1, all_ms.size() * 9 / 10)]);
  return 0;
}

int main(int argc, char **argv) {
  const int n_queries = argc > 1 ? atoi(argv[1]) : 5000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  if (argc > 3 && !strcmp(argv[3], "share")) return bench_share(n_queries, reps);
  if (argc > 3 && !strcmp(argv[3], "and")) return bench_and(n_queries, reps);
  if (argc > 3 && !strcmp(argv[3], "ashare")) return bench_ashare(n_queries, reps);
  const uint32_t max_doc = 10000000u, n_terms = 256;
  std::vector<uint32_t> n_blocks(n_terms);
  for (uint32_t r = 0; r < n_terms; ++r) n_blocks[r] = (max_doc / 2 / (r + 1) + 127) / 128;
  std::mt19937 rng(7);
  std::vector<double> cdf(n_terms);
  double acc = 0;
  for (uint32_t r = 0; r < n_terms; ++r) cdf[r] = (acc += 1.0 / (r + 1));
  PlanScratch ps;
  double best = 1e9;
  uint32_t chunks = 0;
  for (int rep = 0; rep < reps; ++rep) {
    Group &g = ps.groups[1];
    g.reset();
    g.mode = TQ_MODE_OR;
    rng.seed(7);
    for (int q = 0; q < n_queries; ++q) {
      TqdQuery dq{};
      uint32_t picked[5], n = 0;
      while (n < 5) {
        const double u = std::uniform_real_distribution<double>(0, acc)(rng);
        const uint32_t r = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
        bool dup = false;
        for (uint32_t i = 0; i < n; ++i) dup |= picked[i] == r;
        if (!dup) picked[n++] = r;
      }
      std::sort(picked, picked + 5, [](uint32_t a, uint32_t b) { return a > b; });  // rare (heavy) first
      dq.n_terms = 5;
      dq.k = 10;
      dq.flags = TQD_QF_PRUNE;
      uint32_t sparse = 0;
      for (uint32_t i = 0; i < 5; ++i) {
        dq.term[i] = picked[i];
        dq.weight[i] = 2.2f * logf(1.0f + (max_doc - max_doc / 2.0f / (picked[i] + 1)) / (max_doc / 2.0f / (picked[i] + 1)));
        if (picked[i] >= 64) ++sparse;
      }
      const uint32_t c_lb = 1u + 5u + 8u * sparse;
      dq.tile_blocks = std::max<uint32_t>(1u, TQD_AND_TILE * 2u / c_lb);
      uint32_t at = 0;
      for (uint32_t i = 0; i < 5; ++i) {
        dq.lead_tile_start[i] = at;
        at += (n_blocks[picked[i]] + dq.tile_blocks - 1) / dq.tile_blocks;
      }
      for (uint32_t i = 5; i <= TQ_MAX_TERMS; ++i) dq.lead_tile_start[i] = at;
      dq.n_lead = 5;
      dq.n_tiles = at;
      g.queries.push_back(dq);
      g.tile_cost.push_back(dq.tile_blocks * c_lb);
      g.out_index.push_back((uint32_t)q);
      g.max_k = 10;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = build_group_chunks(g, false, ps);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != TQ_OK) return 1;
    best = std::min(best, std::chrono::duration<double, std::milli>(t1 - t0).count());
    chunks = g.n_chunks;
  }
  printf("queries %d chunks %u tiles %u: build_group_chunks %.2f ms (best of %d)\n", n_queries, chunks,
         ps.groups[1].total_tiles, best, reps);
  return 0;
}
