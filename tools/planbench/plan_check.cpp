// Host-only invariants of the chunk planner (build_group_chunks of tq_api.cpp), driven by
// tests/test_planner_cpu.py: synthetic launch groups (candidate unions with per-leader tile runs,
// AND-style single runs), planned with 1 and with N threads.  Checked for every plan:
//   * the chunks tile [0, total_tiles) exactly once, in order;
//   * a chunk's query is the query that holds its first tile;
//   * chunk_recs is a permutation of the chunks (every chunk launched exactly once);
//   * chunk_first / n_parts of a query = the chunks that hold its tiles.
#include "../../tantivy_amd/csrc/tq_api.cpp"

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

int main(int argc, char **argv) {
  const int seed = argc > 1 ? atoi(argv[1]) : 1;
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
